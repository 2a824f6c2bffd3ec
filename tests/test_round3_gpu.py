"""Round-3 features through the C ABI against the CPU oracle: the measured choice of layout and pipeline (frames stay bit-identical while
the renderer probes), the quad replica, the addressing-mode fallback for very long axes, the request-pool proof after an in-place frame,
the degenerate transfer-function range, the asynchronous host mirror of mapframe."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import compare, hip_frame, hip_setup, make_case, oracle_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _render_sequence(ovr, ren, n):
    out = []
    for _ in range(n):
        ren.render()
        st = ren.stats()
        out.append((hip_frame(ovr, ren)[0], st.tuning, st.layout, st.pipeline, (st.samples, st.shaded_samples, st.shadow_samples)))
    return out


@pytest.mark.parametrize("rate", [1.0, 4.0])
def test_measured_choice_keeps_the_frames(ovr, oracle, hip_renderer_factory, rate):
    """A configuration whose shading taps outnumber its primary taps (dense transfer function) makes the renderer time the other pipeline
    and the general / quad layouts, two frames each, and keep the fastest: every frame of the sequence - probe or not - is the frame a
    renderer with OVR_HIP_TUNE=0 produces, bit for bit, and agrees with the oracle; a sparse transfer function stays on the rules."""
    case = make_case(ovr, oracle, n=40, tf="dense", cam="oblique", size=(96, 64), shading=2, rate=rate)
    os.environ["OVR_HIP_TUNE"] = "0"
    try:
        plain = hip_setup(ovr, hip_renderer_factory(), case)
    finally:
        del os.environ["OVR_HIP_TUNE"]
    ref_seq = _render_sequence(ovr, plain, 3)
    assert [t for _, t, *_ in ref_seq] == [0, 0, 0]
    plain.close()
    ren = hip_renderer_factory()
    ren.set_volume_layouts(2)   # all replicas, whatever the free memory
    hip_setup(ovr, ren, case)
    seq = _render_sequence(ovr, ren, 14)
    tun = [t for _, t, *_ in seq]
    assert tun[0] == 0 and tun[1] == 1 and tun[-1] == 2, tun          # rules, probes, the measured winner
    assert 1 in tun and tun.index(2) <= 12, tun                       # at most 1 + 2 * (2 pipelines + 2 more layouts) + 1 frames
    assert {l for _, t, l, *_ in seq if t == 1} >= {0, 3}, seq        # the general and the quad layout were both tried
    assert {p for _, t, _, p, _ in seq if t == 1} == {1, 2}           # ... and both pipelines
    for f, _, _, _, cnt in seq:
        assert np.array_equal(f, ref_seq[0][0]) and cnt == ref_seq[0][4]
    o_rgba, _, cnt = oracle_scene(oracle, case).render()
    assert seq[0][4][0] == cnt.samples and seq[0][4][1] == cnt.shaded_samples
    compare(oracle, seq[-1][0], o_rgba, name=f"tuned rate {rate}")
    # a camera that moves keeps the measured decision (an interactive session never rests long enough to be measured again) ...
    won = (seq[-1][2], seq[-1][3])
    eye, at, up = case["cam"]
    ren.set_camera(ovr.Camera(tuple(c * 1.05 for c in eye), at, up, 60.0))   # still an oblique view: the layout rule says what it said
    ren.commit()
    ren.render()
    st = ren.stats()
    assert st.tuning == 2 and (st.layout, st.pipeline) == won
    # (round 4) ... but a measured LAYOUT only while the layout rule still says what it said when the measurement was made: along an
    # axis the rule asks for a thin replica, and general / quad were never measured against that - the rule's layout, the measured pipeline
    ren.set_camera(ovr.Camera(*ovr.synth.make_camera("front", 40), 60.0))
    ren.commit()
    ren.render()
    st = ren.stats()
    assert st.layout in (1, 2) and st.pipeline == won[1] and st.tuning in (0, 2), (st.layout, st.pipeline, st.tuning)
    later = []
    for _ in range(16):     # ... and measures again once the configuration has rested for a dozen frames
        ren.render()
        later.append(ren.stats().tuning)
    assert later[:10] == [2] * 10 and 1 in later[11:], later
    # any other change starts over with the rules
    colors, alphas, vr = ovr.synth.make_tfn("bumps", 1024, np.float32)
    ren.set_transfer_function(colors, alphas, vr)
    ren.commit()
    ren.render()
    assert ren.stats().tuning == 0
    ren.close()
    # not shade-heavy: the rules stay in charge
    case2 = make_case(ovr, oracle, n=40, tf="sparse", cam="oblique", size=(96, 64), shading=1, rate=1.0)
    ren = hip_setup(ovr, hip_renderer_factory(), case2)
    assert [t for _, t, *_ in _render_sequence(ovr, ren, 4)] == [0, 0, 0, 0]
    ren.close()


def test_measured_choice_under_accumulation(ovr, oracle, hip_renderer_factory):
    """probing must not disturb an accumulation: 12 accumulated frames with and without the measuring are the same frame"""
    case = make_case(ovr, oracle, n=32, tf="dense", cam="front", size=(72, 48), shading=2, rate=2.0)
    frames = []
    for tune in ("0", "1"):
        os.environ["OVR_HIP_TUNE"] = tune
        try:
            ren = hip_renderer_factory()
        finally:
            del os.environ["OVR_HIP_TUNE"]
        ren.set_volume_layouts(2)
        hip_setup(ovr, ren, case, accumulate=True)
        for _ in range(12):
            ren.render()
        frames.append(hip_frame(ovr, ren))
        assert ren.stats().frame_index == 12 and ren.stats().tuning == (2 if tune == "1" else 0)
        ren.close()
    assert np.array_equal(frames[0][0], frames[1][0]) and np.array_equal(frames[0][1], frames[1][1])


def test_very_long_axis_takes_computed_addressing(ovr, oracle, hip_renderer_factory):
    """a volume that is small in bytes but 24 000 voxels long, under a 4096-entry transfer function: the per-axis tables do not
    fit in LDS next to it, the kernels take mode 3 (computed offsets) instead of failing at their first launch"""
    dims = (24000, 6, 5)
    lib = ovr._lib.load()
    assert lib.ovr_hip_query_addressing_mode((C.c_int32 * 3)(*dims), 100, 0, 4096, 4096) == 3
    case = make_case(ovr, oracle, dtype=np.uint8, dims=dims, tf="bumps", size=(80, 40), shading=2, tf_n=4096)
    c = np.array(dims, dtype=np.float64) / 2.0
    case["cam"] = (tuple(c + np.array((300.0, 40.0, 90.0))), tuple(c), (0.0, 1.0, 0.0))
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    ren.render()
    got, _ = hip_frame(ovr, ren)
    st = ren.stats()
    ref, _, cnt = oracle_scene(oracle, case).render()
    assert (st.samples, st.shaded_samples) == (cnt.samples, cnt.shaded_samples) and cnt.samples > 300 and cnt.shaded_samples > 10
    compare(oracle, got, ref, name="long axis")
    ren.close()


def test_an_in_place_frame_does_not_prove_the_pool(ovr, oracle, hip_renderer_factory):
    """an in-place frame followed - with no commit in between - by a pooled frame whose pool is too small: ovr_hip_pack_tiles
    must resolve the frame (it is rendered again with a larger pool) BEFORE packing; the payload is the final frame, stale_tiles stays 0"""
    import torch
    case = make_case(ovr, oracle, n=32, tf="dense", cam="oblique", size=(96, 64), shading=2)
    ref_ren = hip_setup(ovr, hip_renderer_factory(), case)
    ref_ren.render()
    ref, _ = hip_frame(ovr, ref_ren)
    ref_ren.close()
    os.environ["OVR_HIP_POOL_CHUNKS"] = "8"
    os.environ["OVR_HIP_TUNE"] = "0"
    try:
        ren = hip_renderer_factory()
        ren.set_image_shard(0, 1, 16, 16)
        hip_setup(ovr, ren, case, pipeline=1)
        slots = ovr.tiles.max_owned_tiles(96, 64, 16, 16, 1)
        payload = torch.zeros((slots, 16, 16, 4), dtype=torch.float32, device="cuda")
        ren.render()                                   # in place: no pool involved
        assert ren.stats().pipeline == 1
        ren.set_shading_pipeline(2)                    # the same frame, pooled (no accumulation reset, pool never proven)
        ren.commit()
        ren.render_async()
        ovr._lib.check(ren._lib.ovr_hip_pack_tiles(ren._h, C.c_void_p(payload.data_ptr()), payload.numel() * 4))
        ren.sync()
        st = ren.stats()
        assert st.pipeline == 2 and st.pool_chunks > 8 and st.stale_tiles == 0
    finally:
        del os.environ["OVR_HIP_POOL_CHUNKS"]
        del os.environ["OVR_HIP_TUNE"]
    frame = np.zeros((64, 96, 4), np.float32)
    ovr.tiles.unpack_tiles_host(payload.cpu().numpy(), frame, 16, 16, 0, 1)
    assert np.array_equal(frame.reshape(ref.shape), ref)
    ren.close()


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
def test_constant_volume_under_the_data_range_fallback(ovr, oracle, hip_renderer_factory, dtype):
    """a constant volume with the default (invalid) transfer-function range: the data range is one value, the reference's scale is 1 / 0 and its
    coordinate clamp01(0 * inf) = 0 for every sample - the oracle's fminf / fmaxf give that 0, the device gets it from scale = 0"""
    case = make_case(ovr, oracle, n=20, dtype=dtype, tf="dense", cam="oblique", size=(48, 32), shading=2)
    case["vol"] = np.full_like(case["vol"], 7 if np.dtype(dtype) == np.uint8 else 0.25)
    case["vr"] = (1.0, -1.0)
    case["alphas"] = case["alphas"].copy()
    case["alphas"][1] = 0.3          # alpha of entry 0: what every sample must get
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    ren.render()
    got, _ = hip_frame(ovr, ren)
    st = ren.stats()
    ref, _, cnt = oracle_scene(oracle, case).render()
    assert not np.isnan(got).any() and got[..., 3].max() > 0.5
    assert (st.samples, st.shaded_samples, st.shadow_samples) == (cnt.samples, cnt.shaded_samples, cnt.shadow_samples)
    compare(oracle, got, ref, name=f"constant volume {np.dtype(dtype).name}")
    ren.close()


def test_mapframe_copies_only_the_box_rectangle(ovr, oracle, hip_renderer_factory):
    """mapframe(HOST) moves only the pixel rectangle the volume's box projects into; the mirror equals the device frame bit for bit - also
    after the camera moved the rectangle (stale pixels of the old rectangle are refreshed), after a swap, with sparse sampling, and when a
    corner of the box lies behind the camera (whole frame)"""
    import torch
    case = make_case(ovr, oracle, n=32, tf="dense", cam="oblique", size=(200, 120), shading=2)
    ren = hip_setup(ovr, hip_renderer_factory(), case, accumulate=True)

    def check(tag):
        fb_h, fb_d = ovr.FrameBufferData(), ovr.FrameBufferData()
        ren.mapframe(fb_h)
        ren.mapframe(fb_d, device=True)
        for lay in ("rgba", "grad"):
            h = np.array(getattr(fb_h, lay).data(), copy=True)
            d = getattr(fb_d, lay).data().cpu().numpy()
            assert np.array_equal(h.view(np.uint32), d.reshape(h.shape).view(np.uint32)), (tag, lay)
        return np.array(fb_h.rgba.data(), copy=True)

    ren.render(); ren.render()
    a = check("first")
    assert a[..., 3].max() > 0.5 and (a[:, :8] == 0).all() and (a[:, -8:] == 0).all()      # the box does not fill the frame
    eye, at, up = case["cam"]
    far = tuple(np.array(at) + 3.0 * (np.array(eye) - np.array(at)))                           # a smaller rectangle: the old one's border goes back to 0
    ren.set_camera(ovr.Camera(far, at, up, 60.0)); ren.commit(); ren.render()
    b = check("far")
    assert (b[..., 3] > 0).sum() < (a[..., 3] > 0).sum()
    ren.set_camera(ovr.Camera(tuple(np.array(eye) + np.array((40.0, 25.0, 0.0))), tuple(np.array(at) + np.array((40.0, 25.0, 0.0))), up, 60.0)); ren.commit(); ren.render()
    check("shifted")
    ren.swap(); ren.render()
    check("other set")
    # renderapp's order (main_app.cpp:244-263): the NEXT camera is committed before the previous frame is mapped - the mirror must show the frame
    # that is on the device (rendered with the old camera), not the rectangle of the camera just committed
    before = check("before")
    ren.set_camera(ovr.Camera(tuple(np.array(at) + 1.5 * (np.array(eye) - np.array(at)) + np.array((60.0, -30.0, 10.0))), at, up, 60.0)); ren.commit()
    after = check("committed, not rendered")
    assert np.array_equal(before, after)
    ren.render()
    check("rendered")
    ren.set_camera(ovr.Camera(tuple(np.array(at) + 0.2 * (np.array(eye) - np.array(at))), at, up, 60.0)); ren.commit(); ren.render()   # eye inside the box
    c = check("inside")
    assert (c[..., 3] > 0).mean() > 0.9
    ren.set_noise_tile(ovr.synth.make_noise_tile(64)); ren.set_focus((0.5, 0.5), 0.2, 0.1); ren.set_sparse_sampling(True)
    ren.set_camera(ovr.Camera(eye, at, up, 60.0)); ren.commit(); ren.render()
    check("sparse")
    ren.close()


@pytest.mark.parametrize("accumulate", [False, True])
def test_hits_through_an_ignored_slab_reach_the_host_mirror(ovr, oracle, hip_renderer_factory, accumulate):
    """The reference's box test switches a slab off for a ray whose direction component on that axis is below FLT_MIN (shaders_common.h:162-172):
    an axis-aligned camera's centre row hits the box although its rays pass ABOVE it - outside the rectangle the box projects into.
    mapframe(HOST) copies that rectangle only while the march reports no such hit (found by the widened round-3 sweep: seed 11, case 33)."""
    case = make_case(ovr, oracle, n=25, dtype=np.int8, tf="dense", cam="front", size=(59, 17), shading=2, rate=1.0, convention=1, dims=(25, 9, 17),
                     spacing=(1.0, 2.0, 0.5), tf_n=128)
    frames = 3 if accumulate else 1
    ref, _, cnt = oracle_scene(oracle, case).render(frames=frames, accumulate=True)
    ren = hip_setup(ovr, hip_renderer_factory(), case, accumulate=accumulate)

    def frames_of(tag):
        fb_h, fb_d = ovr.FrameBufferData(), ovr.FrameBufferData()
        ren.mapframe(fb_h)
        ren.mapframe(fb_d, device=True)
        h = np.array(fb_h.rgba.data(), copy=True)
        d = fb_d.rgba.data().cpu().numpy().reshape(h.shape)
        assert np.array_equal(h.view(np.uint32), d.view(np.uint32)), tag
        return h

    for _ in range(frames):
        ren.render()
    got = frames_of("axis-aligned")
    compare(oracle, got, ref, name="ignored slab")
    row = got[8, :, 3]                                   # the centre row: sy = 8.5 / 17 = 0.5 exactly, ray direction y == 0
    assert (row > 0).sum() >= 8 and (got[6, :, 3] == 0).all() and (got[10, :, 3] == 0).all()   # ... lit although the rows around it miss the box
    # a camera without such rays: the stale row leaves the mirror again
    eye, at, up = case["cam"]
    ren.set_camera(ovr.Camera(tuple(np.array(eye) + np.array((3.1, -14.3, 2.2))), at, up, 60.0)); ren.commit(); ren.render()
    moved = frames_of("moved")
    assert (ren.stats().frame_index == 1 or not accumulate) and (moved[..., 3] > 0).any()
    ren.close()


def test_device_frame_output_against_the_references_own_conversions(ovr, oracle, hip_renderer_factory):
    """rgba8_kernel and rgba16f_kernel on the WIDE vectors the reference's compiled image_to_rgba8 and save_image("*.exr") + load_exr produced
    (tests/golden/ref_probe_wide.npz): the vectors are written into the device framebuffer, the device converts them, the bits must be the
    reference's - every float within 6 ulp of k / 255, 16 384 floats over the half range with a quarter of them exact ties"""
    import torch
    wide = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_probe_wide.npz"))
    lib = oracle.load()
    for key, size in (("rgba8", (60, 31)), ("exr", (64, 64))):
        vin = wide[key + "_in"].view(np.float32)
        assert vin.size == size[0] * size[1] * 4
        case = make_case(ovr, oracle, n=8, tf="dense", cam="oblique", size=size, shading=0)
        ren = hip_setup(ovr, hip_renderer_factory(), case)
        ren.render()
        fb = ovr.FrameBufferData()
        ren.mapframe(fb, device=True)
        fb.rgba.data().view(-1).copy_(torch.from_numpy(vin.copy()).to(fb.rgba.data().device))
        torch.cuda.synchronize()
        if key == "rgba8":
            got = ren.mapframe_rgba8(flip_vertical=False, device=True).cpu().numpy().reshape(-1)
            assert np.array_equal(got, wide["rgba8_out"])
        else:
            half = ren.mapframe_rgba16f(flip_vertical=False, device=True).cpu().numpy().reshape(-1, 4)
            back = np.array([lib.ovr_oracle_half_to_float(int(v)) for v in half.ravel()], dtype=np.float32).view(np.uint32).reshape(-1, 4)
            assert np.array_equal(back[:, [1, 2, 3, 0]].reshape(-1), wide["exr_out"])
        ren.close()


def test_the_mirror_of_a_frame_rendered_after_an_image_shard(ovr, oracle, hip_renderer_factory):
    """an image shard leaves the tiles of the other ranks as an earlier frame left them - also outside the rectangle the box projects into now;
    mapframe(HOST) must not assume zeros there, and after the shard is switched off the mirror must show the new whole frame
    (found by tests/fuzz_states.py, seeds 61 / 62: host frame != device frame)"""
    case = make_case(ovr, oracle, n=24, tf="dense", cam="inside", size=(45, 43), shading=2)
    ren = hip_setup(ovr, hip_renderer_factory(), case, accumulate=False)

    def mirror_equals_device(tag):
        fb_h, fb_d = ovr.FrameBufferData(), ovr.FrameBufferData()
        ren.mapframe(fb_h)
        ren.mapframe(fb_d, device=True)
        h = np.array(fb_h.rgba.data(), copy=True)
        assert np.array_equal(h.view(np.uint32), fb_d.rgba.data().cpu().numpy().reshape(h.shape).view(np.uint32)), tag
        return h

    ren.render()
    full = mirror_equals_device("camera inside: every pixel lit")
    assert (full[..., 3] > 0).mean() > 0.9
    ren.set_image_shard(0, 2, 16, 8); ren.commit(); ren.render()
    mirror_equals_device("sharded")
    case["cam"] = tuple(ovr.synth.make_camera("front", 24))                      # the box now covers the middle of the frame only
    ren.set_camera(ovr.Camera(*case["cam"], case["fovy"])); ren.commit(); ren.render()
    mirror_equals_device("sharded, small rectangle: the other rank's tiles still hold the first frame")
    ren.set_image_shard(0, 1, 16, 8); ren.commit(); ren.render()
    whole = mirror_equals_device("unsharded again")
    ref, _, _ = oracle_scene(oracle, case).render()
    compare(oracle, whole, ref, name="after the shard")
    assert (whole[:, :4, 3] == 0).all()                                           # the frame's border is outside the box again
    ren.close()


@pytest.mark.parametrize("value", [np.nan, np.inf, -np.inf, 3.0e38, -3.0e38])
def test_non_finite_voxels_inside_the_volume(ovr, oracle, hip_renderer_factory, value):
    """NaN / Inf / huge voxels in a float volume: fmaxf / fminf / clamp behave as in the reference's device build (a NaN sample maps to the lower end
    of the transfer function, a NaN normal to colour 0) - frames and sample counts equal the oracle's, both pipelines.  (Known deviation, DESIGN.md
    section 3: such a voxel in the layer NEXT TO a lower face of the grid leaks into the half-voxel border zone, where the reference's clamp
    addressing reads voxel 0 twice and this repo reads the pair (0, 1) with weight 0 - 0 x NaN; tests/nonfinite_diag.py.)"""
    case = make_case(ovr, oracle, n=14, dtype=np.float32, tf="dense", cam="oblique", size=(48, 40), shading=2, dims=(14, 12, 13), tf_n=128)
    case["vol"][6, 5, 7] = value
    case["vol"][3, 8, 4] = value
    ref, _, cnt = oracle_scene(oracle, case).render()
    for pipeline in (1, 2):
        ren = hip_setup(ovr, hip_renderer_factory(), case, pipeline=pipeline)
        ren.render()
        compare(oracle, hip_frame(ovr, ren)[0], ref, name=f"voxel {value} pipeline {pipeline}")
        assert ren.stats().samples == cnt.samples
        ren.close()
