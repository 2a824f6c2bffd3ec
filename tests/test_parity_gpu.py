"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from helpers import compare, hip_frame, hip_setup, make_case, oracle_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pipeline", [1, 2])
@pytest.mark.parametrize("shading", [0, 1, 2])
@pytest.mark.parametrize("cam", ["front", "oblique"])
@pytest.mark.parametrize("tf", ["sparse", "dense", "bumps"])
def test_frame_parity_f32(ovr, oracle, hip_renderer_factory, shading, cam, tf, pipeline):
    if shading == 0 and pipeline == 2:
        pytest.skip("no shading: there is nothing to pool")
    case = make_case(ovr, oracle, n=32, tf=tf, cam=cam, size=(64, 48), shading=shading)
    ref_rgba, ref_grad, cnt = oracle_scene(oracle, case).render()
    ren = hip_setup(ovr, hip_renderer_factory(), case, pipeline=pipeline)
    ren.render()
    assert ren.stats().pipeline == (pipeline if shading else 1)
    rgba, grad = hip_frame(ovr, ren)
    compare(oracle, rgba, ref_rgba, name=f"{tf}/{cam}/{shading}")
    if shading:
        assert np.abs(grad - ref_grad).max() <= 2e-3
    st = ren.stats()
    assert st.rays == cnt.rays
    assert st.samples == cnt.samples, "primary sample count differs from the oracle"
    assert st.shaded_samples == cnt.shaded_samples
    if shading == 2:
        assert st.shadow_samples == cnt.shadow_samples_visible


@pytest.mark.parametrize("spp", [1, 4])
@pytest.mark.parametrize("shading", [1, 2])
def test_pipelines_bit_identical(ovr, oracle, hip_renderer_factory, shading, spp):
    """in-place and pooled shading apply every pixel's contributions in the same order: the frames must be equal bit for bit
    (spp > 1: the pooled pipeline runs once per sample-per-pixel generation and sums the generations in order)"""
    case = make_case(ovr, oracle, n=48, tf="bumps", cam="oblique", size=(160, 96), shading=shading, spp=spp)
    frames = []
    for pipeline in (1, 2):
        ren = hip_setup(ovr, hip_renderer_factory(), case, pipeline=pipeline)
        ren.render()
        frames.append(hip_frame(ovr, ren) + (ren.stats(),))
    assert np.array_equal(frames[0][0], frames[1][0])
    assert np.array_equal(frames[0][1], frames[1][1])
    for k in ("rays", "samples", "shaded_samples", "shadow_samples"):
        assert getattr(frames[0][2], k) == getattr(frames[1][2], k)
    assert frames[1][2].pool_chunks > 0


@pytest.mark.parametrize("dtype", [np.uint8, np.int8, np.uint16, np.int16, np.float64, np.uint32, np.int32])
def test_dtypes(ovr, oracle, hip_renderer_factory, dtype):
    """every reference ValueType: u8/i8 normalised reads, u16/i16/f64 sampled as raw float (array.cpp:322-347)"""
    case = make_case(ovr, oracle, n=24, dtype=dtype, tf="bumps", cam="oblique", size=(48, 40), shading=2)
    if dtype in (np.uint32, np.int32):
        # normalised 32-bit reads: the TF range is given in raw units and normalised like the data (array.h:96-99)
        # (float cannot hold UINT32_MAX / INT32_MAX: the casts in integer_normalize would overflow, in the reference too)
        case["vr"] = (0.0, 4.0e9) if dtype == np.uint32 else (-2.0e9, 2.0e9)
    ref_rgba, ref_grad, cnt = oracle_scene(oracle, case).render()
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    ren.render()
    rgba, grad = hip_frame(ovr, ren)
    # integer volumes have flat regions: zero gradients -> NaN normals -> colour 0 on both sides
    compare(oracle, rgba, ref_rgba, tol_float=5e-4, name=str(dtype))
    st = ren.stats()
    assert st.samples == cnt.samples and st.shaded_samples == cnt.shaded_samples
    assert st.shadow_samples == cnt.shadow_samples_visible
    assert cnt.shaded_samples > 0


@pytest.mark.parametrize("pipeline", [1, 2])
def test_non_cubic_spacing_origin_vertex_convention(ovr, oracle, hip_renderer_factory, pipeline):
    case = make_case(ovr, oracle, n=0, dims=(40, 23, 31), tf="bumps", cam="oblique", size=(57, 43), shading=2, convention=1,
                     spacing=(1.0, 1.5, 0.75), origin=(3.0, -2.0, 5.0), rate=2.0)
    ref_rgba, ref_grad, cnt = oracle_scene(oracle, case).render()
    ren = hip_setup(ovr, hip_renderer_factory(), case, pipeline=pipeline)
    ren.render()
    rgba, grad = hip_frame(ovr, ren)
    compare(oracle, rgba, ref_rgba, name="aniso")
    assert np.abs(grad - ref_grad).max() <= 2e-3
    st = ren.stats()
    assert (st.rays, st.samples, st.shaded_samples) == (cnt.rays, cnt.samples, cnt.shaded_samples)
    assert st.active_pixels == 57 * 43


def test_camera_inside_volume(ovr, oracle, hip_renderer_factory):
    case = make_case(ovr, oracle, n=32, tf="sparse", cam="inside", size=(64, 64), shading=2)
    ref_rgba, _, cnt = oracle_scene(oracle, case).render()
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    ren.render()
    rgba, _ = hip_frame(ovr, ren)
    compare(oracle, rgba, ref_rgba, name="inside")
    assert ren.stats().samples == cnt.samples


@pytest.mark.parametrize("spp", [1, 3])
def test_accumulation_and_spp(ovr, oracle, hip_renderer_factory, spp):
    """frame accumulation over 3 frames (shaders_raymarching.cu:389-403) and TEA pixel jitter when spp > 1 (:351-357)"""
    case = make_case(ovr, oracle, n=24, tf="bumps", cam="oblique", size=(40, 32), shading=2, spp=spp)
    ref_rgba, _, cnt = oracle_scene(oracle, case).render(frames=3, accumulate=True)
    ren = hip_setup(ovr, hip_renderer_factory(), case, accumulate=True)
    for _ in range(3):
        ren.render()
    st = ren.stats()
    assert st.frame_index == 3
    rgba, _ = hip_frame(ovr, ren)
    compare(oracle, rgba, ref_rgba, name=f"accum spp{spp}")
    assert st.rays == 40 * 32 * spp and st.samples == cnt.samples
    # any committed change restarts the accumulation (device_impl.cpp:116-196,225-233)
    ren.set_volume_sampling_rate(1.0)
    ren.commit()
    ren.render()
    assert ren.stats().frame_index == 1


def test_sparse_sampling_mask_and_frame(ovr, oracle, hip_renderer_factory):
    """foveated sparse sampling: the compacted pixel list must be bit-identical (integer work), the frame matches at the
    sampled pixels and is zero elsewhere (device_impl.cpp:234-239)"""
    rng = np.random.default_rng(11)
    noise = (rng.integers(0, 256, size=(64, 64, 64)) / 255.0).astype(np.float32)
    case = make_case(ovr, oracle, n=24, tf="bumps", cam="oblique", size=(72, 40), shading=2)
    focus = ((0.45, 0.55), 0.25, 0.1)
    sc = oracle_scene(oracle, case, sparse=True, focus=focus, noise=noise)
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    ren.set_noise_tile(noise)
    ren.set_focus(*focus)
    ren.set_sparse_sampling(True)
    ren.commit()
    for frame in (1, 2, 70):
        exp = oracle.sparse_mask(frame, 72, 40, focus[0], focus[1], focus[2], noise)
        got = ren.sparse_mask(frame)
        assert np.array_equal(got, exp), f"mask differs at frame {frame}"
    ren.render()   # frame_index 1, no accumulation
    rgba, _ = hip_frame(ovr, ren)
    ref_rgba, _, cnt = sc.render(frames=1)
    compare(oracle, rgba, ref_rgba, name="sparse frame")
    st = ren.stats()
    n_kept = len(oracle.sparse_mask(1, 72, 40, focus[0], focus[1], focus[2], noise)) // 2
    assert st.active_pixels == n_kept and 0 < n_kept < 72 * 40
    assert st.samples == cnt.samples


def test_tea_kat_on_device(ovr, oracle, hip_renderer_factory):
    ren = hip_setup(ovr, hip_renderer_factory(), make_case(ovr, oracle, n=8, size=(8, 8)))
    seeds = np.array([[1, 0], [1, 12345], [7, 2073599], [0xFFFFFFFF, 0xFFFFFFFF], [0, 0]], dtype=np.uint32)
    floats, states = ren.tea_floats(seeds)
    for i, (a, b) in enumerate(seeds):
        (f0, f1), st = oracle.tea_floats(int(a), int(b))
        assert floats[2 * i] == np.float32(f0) and floats[2 * i + 1] == np.float32(f1)
        assert tuple(states.reshape(-1, 2)[i]) == st


def test_double_buffer_and_device_mapping(ovr, oracle, hip_renderer_factory):
    """renderapp's protocol: commit, mapframe (previous frame), swap, render (apps/main_app.cpp:244-263); the mapped frame
    must stay intact while the next one renders into the other set; device mapping hands out HBM without a copy"""
    import torch
    case = make_case(ovr, oracle, n=24, tf="bumps", cam="front", size=(48, 32), shading=1)
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    ren.render()
    fb_dev = ovr.FrameBufferData()
    ren.mapframe(fb_dev, device=True)
    assert fb_dev.rgba.data().is_cuda and fb_dev.rgba.data().shape == (32, 48, 4)
    first = fb_dev.rgba.to_cpu().data().copy()
    ren.swap()
    eye, at, up = case["cam"]
    ren.set_camera((eye[0] + 5.0, eye[1], eye[2]), at, up)
    ren.commit()
    ren.render()
    torch.cuda.synchronize()
    assert np.array_equal(fb_dev.rgba.to_cpu().data(), first), "the mapped set was overwritten by the next render"
    second, _ = hip_frame(ovr, ren)
    assert not np.array_equal(second, first)


def test_image_shards_reassemble_bit_exactly(ovr, oracle, hip_renderer_factory):
    """multi-GPU image-plane sharding on ONE card: 3 renderers play 3 ranks; pack -> unpack must rebuild the unsharded frame"""
    import ctypes as C
    import torch
    case = make_case(ovr, oracle, n=24, tf="bumps", cam="oblique", size=(80, 56), shading=2)
    full = hip_setup(ovr, hip_renderer_factory(), case)
    full.render()
    ref, _ = hip_frame(ovr, full)
    world, TW, TH = 3, 16, 8
    slots = ovr.tiles.max_owned_tiles(80, 56, TW, TH, world)
    frame = torch.zeros((56, 80, 4), dtype=torch.float32, device="cuda")
    root = None
    for rank in range(world):
        ren = hip_renderer_factory()
        ren.set_image_shard(rank, world, TW, TH)
        hip_setup(ovr, ren, case)
        ren.render()
        assert ren.stats().active_pixels == sum(min(TW, 80 - tx * TW) * min(TH, 56 - ty * TH) for tx, ty in ovr.tiles.owned_tiles(80, 56, TW, TH, rank, world))
        payload = torch.zeros((slots, TH, TW, 4), dtype=torch.float32, device="cuda")
        ovr._lib.check(ren._lib.ovr_hip_pack_tiles(ren._h, C.c_void_p(payload.data_ptr()), payload.numel() * 4))
        ren.sync()
        root = root or ren
        torch.cuda.synchronize()
        ovr._lib.check(root._lib.ovr_hip_unpack_tiles(root._h, rank, C.c_void_p(payload.data_ptr()), payload.numel() * 4,
                                                      C.c_void_p(frame.data_ptr()), frame.numel() * 4))
        root.sync()
    torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy(), ref)


def test_errors_are_loud(ovr, oracle, hip_renderer_factory):
    ren = hip_renderer_factory()
    ren.set_fbsize((16, 16))
    ren.commit()
    with pytest.raises(RuntimeError, match="before a volume was set"):
        ren.render()
    with pytest.raises(RuntimeError, match="positive"):
        ren.set_sample_per_pixel(0)
    with pytest.raises(RuntimeError, match="path tracing"):
        ren.set_path_tracing(True)


@pytest.mark.parametrize("dtype", [np.float32, np.uint8, np.uint16])
def test_macrocell_grids_match_oracle(ovr, oracle, hip_renderer_factory, dtype):
    """value-range and majorant grids of the reference's SpacePartiton_SingleMC (sp_singlemc.cu:10-97)"""
    case = make_case(ovr, oracle, n=0, dims=(40, 33, 50), dtype=dtype, tf="bumps", cam="oblique", size=(32, 32), shading=2)
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    mm, mj = ren.macrocells()
    rmm, rmj = oracle_scene(oracle, case).macrocells()
    assert mm.shape == (4, 3, 3, 2)
    assert np.array_equal(mm, rmm)
    assert np.array_equal(mj, rmj)
    assert (mj == 0).any() and (mj > 0).any()


@pytest.mark.parametrize("pipeline", [1, 2])
@pytest.mark.parametrize("tf", ["sparse", "bumps"])
def test_empty_space_skipping_is_bit_identical(ovr, oracle, hip_renderer_factory, tf, pipeline):
    case = make_case(ovr, oracle, n=64, tf=tf, cam="oblique", size=(96, 72), shading=2)
    frames = []
    for skip in (False, True):
        ren = hip_setup(ovr, hip_renderer_factory(), case, pipeline=pipeline)
        ren.set_empty_space_skipping(skip)
        ren.commit()
        ren.render()
        frames.append(hip_frame(ovr, ren) + (ren.stats(),))
    (a_rgba, a_grad, a), (b_rgba, b_grad, b) = frames
    assert np.array_equal(a_rgba, b_rgba) and np.array_equal(a_grad, b_grad)
    assert a.skipped_samples == 0 and b.skipped_samples > 0
    assert a.samples == b.samples + b.skipped_samples
    assert a.shaded_samples == b.shaded_samples
    assert a.shadow_samples == b.shadow_samples + b.skipped_shadow_samples
    _, _, cnt = oracle_scene(oracle, case).render()
    assert cnt.samples == b.samples + b.skipped_samples


@pytest.mark.parametrize("dims,size", [((1, 1, 1), (9, 7)), ((2, 3, 5), (13, 7)), ((31, 30, 29), (1, 1)), ((33, 5, 70), (25, 41))])
def test_ragged_sizes(ovr, oracle, hip_renderer_factory, dims, size):
    """grids that are not multiples of the brick / macro block / macrocell sizes, framebuffers that are not multiples of the
    4x4 ray tile, single voxels and single pixels"""
    case = make_case(ovr, oracle, n=0, dims=dims, tf="dense", cam="oblique", size=size, shading=2)
    ref_rgba, ref_grad, cnt = oracle_scene(oracle, case).render()
    for skip in (False, True):
        ren = hip_setup(ovr, hip_renderer_factory(), case)
        ren.set_empty_space_skipping(skip)
        ren.commit()
        ren.render()
        rgba, grad = hip_frame(ovr, ren)
        compare(oracle, rgba, ref_rgba, name=f"{dims} {size}")
        st = ren.stats()
        assert st.samples + st.skipped_samples == cnt.samples and st.active_pixels == size[0] * size[1]


def test_large_transfer_function_and_tiny_one(ovr, oracle, hip_renderer_factory):
    for tf_n in (2, 4096):
        case = make_case(ovr, oracle, n=24, tf="dense", cam="front", size=(40, 32), shading=2, tf_n=tf_n)
        ref_rgba, _, cnt = oracle_scene(oracle, case).render()
        ren = hip_setup(ovr, hip_renderer_factory(), case)
        ren.render()
        rgba, _ = hip_frame(ovr, ren)
        compare(oracle, rgba, ref_rgba, name=f"tf {tf_n}")
        assert ren.stats().samples == cnt.samples
    # beyond what fits in LDS the backend refuses loudly instead of silently degrading
    case = make_case(ovr, oracle, n=8, tf="dense", cam="front", size=(8, 8), shading=2, tf_n=8192)
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    with pytest.raises(RuntimeError, match="transfer function too large"):
        ren.render()


def test_pool_overflow_is_recovered(ovr, oracle, hip_renderer_factory, monkeypatch):
    """the request pool starts too small (forced with OVR_HIP_POOL_CHUNKS): the march overflows it, the frame is re-rendered with
    a larger pool and must still be exact (accumulation must not see the aborted attempt)"""
    monkeypatch.setenv("OVR_HIP_POOL_CHUNKS", "64")
    case = make_case(ovr, oracle, n=48, tf="dense", cam="oblique", size=(64, 64), shading=2, rate=4.0)
    ref_rgba, _, cnt = oracle_scene(oracle, case).render(frames=2, accumulate=True)
    ren = hip_setup(ovr, hip_renderer_factory(), case, accumulate=True, pipeline=2)
    ren.render()
    ren.render()
    rgba, _ = hip_frame(ovr, ren)
    compare(oracle, rgba, ref_rgba, name="overflow")
    st = ren.stats()
    assert st.frame_index == 2 and st.shaded_samples == cnt.shaded_samples and st.pool_chunks > 64


def test_render_from_scene_file(tmp_path, ovr, oracle, hip_renderer_factory):
    """f1 end to end in Python: VIDI3D file -> our loader -> DeviceHIP, with the scene's own fovy and sampling rate"""
    vol = ovr.synth.make_volume(32, np.uint8)
    colors, alphas, vr = ovr.synth.make_tfn("bumps", 256, np.uint8)
    cam = ovr.synth.make_camera("oblique", 32)
    path = ovr.vidi3d.write_scene(str(tmp_path), "s", vol, ovr.synth._RAINBOW, alphas[1::2], (0.0, 1.0), cam, fovy=45.0, sample_distance=0.5)
    scene, camera = ovr.vidi3d.scene_from_file(path)
    ren = hip_renderer_factory()
    ren.set_fbsize((64, 48))
    ren.init(scene, camera)
    ren.render()
    rgba, _ = hip_frame(ovr, ren)
    d = ovr.vidi3d.read_scene(path)
    tfc = d["tfn_color"][:, :3].ravel()
    tfa = np.stack([np.linspace(0, 1, 256, dtype=np.float32), d["tfn_opacity"]], 1).ravel()
    ref, _, cnt = oracle.OracleScene(vol, tfc, tfa, d["value_range"], cam, 64, 48, fovy=45.0, rate=2.0, shading=oracle.SHADE_FULL).render()
    compare(oracle, rgba, ref, name="scene file")
    assert ren.stats().samples == cnt.samples


@pytest.mark.parametrize("xy,size", [(128, (300, 21)), (64, (640, 5)), (128, (97, 33))])
def test_sparse_mask_tile_sizes(ovr, oracle, hip_renderer_factory, xy, size):
    """STBN-sized (128) and blue-noise-sized (64) tiles, rows wider and narrower than a workgroup's 256 pixels"""
    rng = np.random.default_rng(xy)
    noise = (rng.integers(0, 256, size=(xy, xy, 64)) / 255.0).astype(np.float32)
    case = make_case(ovr, oracle, n=8, tf="dense", cam="front", size=size, shading=0)
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    ren.set_noise_tile(noise)
    focus = ((0.3, 0.6), 0.4, 0.05)
    ren.set_focus(*focus)
    ren.commit()
    for frame in (1, 63, 64, 129):
        assert np.array_equal(ren.sparse_mask(frame), oracle.sparse_mask(frame, size[0], size[1], focus[0], focus[1], focus[2], noise))


@pytest.mark.parametrize("dtype", [np.float32, np.uint16, np.uint8])
def test_addressing_modes_bit_identical(ovr, oracle, hip_renderer_factory, monkeypatch, dtype):
    """the four addressing modes of the brick fetch (32-bit byte offsets / 32-bit element offsets / 64-bit z table in LDS /
    64-bit computed) are chosen by volume size; forced on one small volume (OVR_HIP_ADDRESSING) they must agree bit for bit, in
    both pipelines and with skipping"""
    case = make_case(ovr, oracle, n=40, tf="bumps", cam="oblique", size=(72, 56), shading=2, dtype=dtype)
    frames = []
    for am in (0, 1, 2, 3):
        monkeypatch.setenv("OVR_HIP_ADDRESSING", str(am))
        for pipeline, skip in ((2, False), (1, False), (2, True)):
            ren = hip_setup(ovr, hip_renderer_factory(), case, pipeline=pipeline)
            ren.set_empty_space_skipping(skip)
            ren.commit()
            ren.render()
            frames.append(hip_frame(ovr, ren)[0].copy())
            ren.close()
    for f in frames[1:]:
        assert np.array_equal(f, frames[0])
    ref, _, _ = oracle_scene(oracle, case).render()
    compare(oracle, frames[0], ref, name="addressing")


def test_mapframe_rgba8_equals_reference_conversion(ovr, oracle, hip_renderer_factory):
    """f4 frame output on the device: ovr_hip_mapframe_rgba8 == image_to_rgba8 (the oracle's restatement is pinned bit-exactly
    against the reference's compiled function, tests/test_oracle_vs_ref.py), flipped and not, host and device memory"""
    case = make_case(ovr, oracle, n=32, tf="dense", cam="oblique", size=(70, 45), shading=2)
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    ren.render()
    rgba, _ = hip_frame(ovr, ren)
    assert rgba.max() > 0.5
    for flip in (True, False):
        want = oracle.rgba8(rgba, flip=flip)
        got = np.array(ren.mapframe_rgba8(flip_vertical=flip), copy=True)
        assert got.shape == (45, 70, 4) and got.dtype == np.uint8
        assert np.array_equal(got.reshape(want.shape), want)
        dev = ren.mapframe_rgba8(flip_vertical=flip, device=True)
        assert np.array_equal(dev.cpu().numpy().reshape(want.shape), want)
    # save_image: what renderbatch's ovr::save_image writes for this frame (flipped rows)
    from PIL import Image
    import os, tempfile
    with tempfile.TemporaryDirectory() as d:
        ren.save_image(os.path.join(d, "frame.png"))
        png = np.asarray(Image.open(os.path.join(d, "frame.png")).convert("RGBA"))
    assert np.array_equal(png, oracle.rgba8(rgba, flip=True))


def test_state_changes_on_one_renderer(ovr, oracle, hip_renderer_factory):
    """one renderer through the transitions an interactive app makes - camera move, resize, transfer-function edit, shading
    and pipeline switch, shard on and off, new volume - every frame against the oracle.  Guards the commit logic
    (device_impl.cpp:113-197) and everything cached per camera / size / shard (the workgroup schedule)."""
    case = make_case(ovr, oracle, n=36, tf="bumps", cam="oblique", size=(88, 56), shading=2)
    ren = hip_setup(ovr, hip_renderer_factory(), case, accumulate=True)

    def check(name, frames=1, **kw):
        for _ in range(frames):
            ren.render()
        ref, _, cnt = oracle_scene(oracle, case, **kw).render(frames=frames, accumulate=True)
        got = hip_frame(ovr, ren)[0]
        st = ren.stats()
        assert st.frame_index == frames, name
        assert st.samples == cnt.samples, name
        return got, ref

    compare(oracle, *check("initial", frames=2), name="initial")
    # camera move: accumulation restarts, the schedule is re-sorted
    case["cam"] = tuple(ovr.synth.make_camera("front", 36))
    ren.set_camera(ovr.Camera(*case["cam"], case["fovy"]))
    ren.commit()
    compare(oracle, *check("camera"), name="camera")
    # resize (odd size: partial 8x8 blocks on both edges)
    case["size"] = (61, 83)
    ren.set_fbsize(case["size"])
    ren.commit()
    compare(oracle, *check("resize", frames=2), name="resize")
    # transfer-function edit and shading mode
    case["colors"], case["alphas"], case["vr"] = ovr.synth.make_tfn("dense", 64)
    case["shading"] = 1
    ren.set_transfer_function(case["colors"], case["alphas"], case["vr"])
    ren.set_shading(1)
    ren.set_shading_pipeline(1)
    ren.commit()
    compare(oracle, *check("tfn+shading"), name="tfn+shading")
    # shard on: only the owned tiles are drawn (the others keep what they had), then off again
    ren.set_shading_pipeline(2)
    ren.set_image_shard(1, 3, 16, 8)
    ren.commit()
    got, ref = check("shard", shard=(1, 3, 16, 8))
    mask = np.zeros(got.shape[:2], bool)
    for tx, ty in ovr.tiles.owned_tiles(case["size"][0], case["size"][1], 16, 8, 1, 3):
        mask[ty * 8:(ty + 1) * 8, tx * 16:(tx + 1) * 16] = True
    compare(oracle, np.where(mask[..., None], got, 0.0).astype(np.float32), np.where(mask[..., None], ref, 0.0).astype(np.float32), name="shard")
    ren.set_image_shard(0, 1, 16, 8)
    ren.commit()
    compare(oracle, *check("unshard"), name="unshard")
    # a new volume of another shape and type on the same renderer
    case.update(make_case(ovr, oracle, n=24, dtype=np.uint8, tf="bumps", cam="oblique", size=case["size"], shading=1, dims=(40, 24, 31)))
    ren.set_transfer_function(case["colors"], case["alphas"], case["vr"])
    eye, at, up = case["cam"]
    ren.init(ovr.Scene(volume=case["vol"], transfer_function=None), ovr.Camera(eye, at, up, case["fovy"]))
    ren.commit()
    compare(oracle, *check("volume"), name="volume")
