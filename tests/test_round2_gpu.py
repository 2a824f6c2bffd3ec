"""Round-2 features through the C ABI against the CPU oracle: the data-range fallback of an invalid transfer-function range
(volume.cpp:131-145, array.cpp:27-66,297), blue-noise pixel jitter (BASELINE C5), the EXR half frame (imageio.cpp:15-83), the
prologue/march agreement on frame widths that are not powers of two, and the pool-overflow path under the pipelined gather."""
import ctypes as C
import os
import subprocess
import sys
import json

import numpy as np
import pytest

from helpers import compare, hip_frame, hip_setup, make_case, oracle_scene, run_bench

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dtype", [np.float32, np.uint8, np.int8, np.uint16, np.int16, np.float64, np.uint32])
def test_data_range_fallback(ovr, oracle, hip_renderer_factory, dtype):
    """the default TransferFunction range (1, -1) is invalid: the data range found at load stays in effect"""
    case = make_case(ovr, oracle, n=24, dtype=dtype, cam="oblique", size=(56, 40))
    case["vr"] = (1.0, -1.0)
    sc = oracle_scene(oracle, case)
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    info = ren.volume_info()
    lo, hi = sc.data_range()
    assert (np.float32(info.data_lower), np.float32(info.data_upper)) == (np.float32(lo), np.float32(hi))
    assert (np.float32(info.tf_lower), np.float32(info.tf_upper)) == (np.float32(lo), np.float32(hi))
    assert tuple(info.dims) == (24, 24, 24) and info.resident_bytes >= case["vol"].size * min(case["vol"].itemsize, 4)
    ren.render()
    got, _ = hip_frame(ovr, ren)
    ref, _, cnt = sc.render()
    st = ren.stats()
    assert (st.samples, st.shaded_samples) == (cnt.samples, cnt.shaded_samples) and cnt.shaded_samples > 0
    compare(oracle, got, ref, name=f"fallback {np.dtype(dtype).name}")
    # a valid range replaces it, a later invalid one keeps the valid one (set_value_range only overwrites on hi >= lo)
    vr2 = (float(case["vol"].min()) * 0.5 + float(case["vol"].max()) * 0.5, float(case["vol"].max()))
    ren.set_transfer_function(case["colors"], case["alphas"], vr2)
    ren.commit()
    i2 = ren.volume_info()
    assert i2.tf_lower > info.tf_lower and i2.data_lower == info.data_lower
    ren.render()
    got2, _ = hip_frame(ovr, ren)
    ref2, _, _ = oracle_scene(oracle, dict(case, vr=vr2)).render()
    compare(oracle, got2, ref2, name="valid range")
    ren.set_transfer_function(case["colors"], case["alphas"], (1.0, -1.0))
    ren.commit()
    i3 = ren.volume_info()
    assert (i3.tf_lower, i3.tf_upper) == (i2.tf_lower, i2.tf_upper)
    ren.close()


def test_data_range_fallback_with_skipping_and_macrocells(ovr, oracle, hip_renderer_factory):
    case = make_case(ovr, oracle, n=40, cam="oblique", size=(64, 48))
    case["vr"] = (1.0, -1.0)
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    ren.render()
    plain, _ = hip_frame(ovr, ren)
    st0 = ren.stats()
    ren.set_empty_space_skipping(True)
    ren.commit()
    ren.render()
    skip, _ = hip_frame(ovr, ren)
    st1 = ren.stats()
    assert np.array_equal(plain, skip) and st1.samples + st1.skipped_samples == st0.samples
    mm, mj = ren.macrocells()
    omm, omj = oracle_scene(oracle, case).macrocells()
    assert np.array_equal(mm, omm) and np.array_equal(mj, omj)
    ren.close()


@pytest.mark.parametrize("spp,pipeline,skip,shading", [(1, 0, False, 2), (1, 1, False, 2), (4, 2, False, 2), (4, 1, False, 2), (3, 0, True, 2),
                                                     (1, 0, True, 1), (6, 0, False, 0), (2, 0, False, 1)])
def test_blue_noise_jitter_vs_oracle(ovr, oracle, hip_renderer_factory, spp, pipeline, skip, shading):
    """pixel jitter from the noise tile, slice ((frame - 1) * spp + k) % 64: three accumulated frames against the oracle; more
    samples per pixel than the kernel stages in LDS (6 > 4) take the direct lookup"""
    noise = ovr.synth.make_noise_tile(16 if spp != 4 else 64, seed=11)
    case = make_case(ovr, oracle, n=28, tf="bumps", cam="oblique", size=(72, 40), spp=spp, shading=shading)
    ren = hip_renderer_factory()
    ren.set_noise_tile(noise)
    ren.set_pixel_jitter(ovr.JITTER_BLUE_NOISE)
    ren.set_empty_space_skipping(skip)
    hip_setup(ovr, ren, case, accumulate=True, pipeline=pipeline)
    sc = oracle_scene(oracle, case, jitter=1, noise=noise)
    rendered = 0
    for frames in (1, 3):
        while rendered < frames:
            ren.render()
            rendered += 1
        got, _ = hip_frame(ovr, ren)
        ref, _, cnt = sc.render(frames=frames, accumulate=True)
        st = ren.stats()
        assert st.frame_index == frames
        assert st.samples + st.skipped_samples == cnt.samples and st.shaded_samples == cnt.shaded_samples
        compare(oracle, got, ref, name=f"jitter spp {spp} frames {frames}")
    # the jittered frame differs from the unjittered one, and TEA mode is back after the switch
    ren.set_pixel_jitter(ovr.JITTER_TEA)
    ren.commit()
    ren.render()
    tea, _ = hip_frame(ovr, ren)
    ref_tea, _, _ = oracle_scene(oracle, case).render(frames=1, accumulate=True)
    compare(oracle, tea, ref_tea, name="back to TEA")
    assert not np.array_equal(tea, got)
    ren.close()


def test_blue_noise_jitter_sparse_and_sharded(ovr, oracle, hip_renderer_factory):
    noise = ovr.synth.make_noise_tile(16, seed=5)
    case = make_case(ovr, oracle, n=24, cam="oblique", size=(80, 48), spp=2)
    focus = ((0.5, 0.45), 0.3, 0.15)
    ren = hip_renderer_factory()
    ren.set_noise_tile(noise)
    ren.set_pixel_jitter(1)
    ren.set_focus(*focus)
    ren.set_image_shard(1, 3, 16, 16)
    hip_setup(ovr, ren, case)
    ren.set_sparse_sampling(True)
    ren.commit()
    ren.render()
    got, _ = hip_frame(ovr, ren)
    sc = oracle_scene(oracle, case, jitter=1, noise=noise, sparse=True, focus=focus, shard=(1, 3, 16, 16))
    ref, _, cnt = sc.render()
    assert ren.stats().samples == cnt.samples
    compare(oracle, got, ref, name="jitter sparse shard")
    ren.close()


def test_jitter_needs_a_noise_tile(ovr, hip_renderer_factory, oracle):
    case = make_case(ovr, oracle, n=8, size=(16, 16))
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    ren.set_pixel_jitter(1)
    ren.commit()
    with pytest.raises(RuntimeError, match="noise tile"):
        ren.render()
    with pytest.raises(RuntimeError, match="jitter"):
        ren.set_pixel_jitter(7)
    ren.close()


@pytest.mark.parametrize("size", [(1920, 8), (1366, 24), (1000, 16), (333, 77)])
def test_prologue_and_march_agree_on_any_width(ovr, oracle, hip_renderer_factory, size):
    """frame widths that are not powers of two, camera placed so that the volume's silhouette crosses many 8x8 blocks: the
    workgroup's staging decision and the march use the same ray ((ix + .5) / W vs (ix + .5) * (1 / W))"""
    case = make_case(ovr, oracle, n=16, tf="dense", cam="oblique", size=size, shading=1, fovy=25.0)
    for skip in (False, True):
        ren = hip_renderer_factory()
        ren.set_empty_space_skipping(skip)
        hip_setup(ovr, ren, case)
        ren.render()
        got, _ = hip_frame(ovr, ren)
        ref, _, cnt = oracle_scene(oracle, case).render()
        st = ren.stats()
        assert st.samples + st.skipped_samples == cnt.samples
        compare(oracle, got, ref, name=f"width {size} skip {skip}")
        ren.close()


def test_exr_half_frame_and_files(ovr, oracle, hip_renderer_factory, tmp_path):
    """the half frame of the EXR writer: bit-exact against the oracle's restatement of tinyexr's conversion, flipped like
    save_image flips (imageio.cpp:271); the files save_image writes decode to the same pixels"""
    case = make_case(ovr, oracle, n=24, tf="dense", cam="oblique", size=(96, 56), shading=2)
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    ren.render()
    rgba, _ = hip_frame(ovr, ren)
    # inject values that exercise the rounding rule into a copy of the frame on the device
    import torch
    fb = ovr.FrameBufferData()
    ren.mapframe(fb, device=True)
    t = fb.rgba.data()
    special = torch.tensor([1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 65520.0, 1e9, -1e9, 2.0 ** -24, 2.0 ** -25, 2.0 ** -26, 1e-45, -0.0,
                            float("inf"), float("nan"), 2.0 - 2.0 ** -12, 6.1e-5, 6.0e-5, 0.1], dtype=torch.float32, device=t.device)
    t.view(-1)[: special.numel()] = special
    rgba = t.cpu().numpy().copy()
    exp = oracle.float_to_half(rgba)
    for flip in (False, True):
        got = np.array(ren.mapframe_rgba16f(flip_vertical=flip), copy=True)
        assert got.dtype == np.uint16 and got.shape == (56, 96, 4)
        e = exp[::-1] if flip else exp
        assert np.array_equal(got, e), np.argwhere(got != e)[:5]
    dev = ren.mapframe_rgba16f(flip_vertical=True, device=True)
    assert np.array_equal(dev.cpu().numpy().view(np.uint16), exp[::-1])
    # files
    t.view(-1)[: special.numel()] = 0.25   # finite pixels for the file round trips
    ren.save_image(str(tmp_path / "f.exr"))
    ren.save_image(str(tmp_path / "f.png"))
    ren.save_image(str(tmp_path / "f.jpg"))
    from PIL import Image
    rgba8 = np.array(ren.mapframe_rgba8(flip_vertical=True), copy=True)
    assert np.array_equal(np.array(Image.open(tmp_path / "f.png")), rgba8)
    jpg = np.array(Image.open(tmp_path / "f.jpg")).astype(int)
    assert jpg.shape == (56, 96, 3) and np.abs(jpg - rgba8[..., :3].astype(int)).mean() < 3.0
    raw = open(tmp_path / "f.exr", "rb").read()
    assert raw[:4] == bytes([0x76, 0x2f, 0x31, 0x01]) and b"channels\0chlist\0" in raw and b"compression\0compression\0" in raw
    ren.close()


def test_pool_overflow_under_the_pipelined_gather(ovr, oracle, hip_renderer_factory):
    """with a request pool that overflows, the frame is rendered again inside the host wait - the tiles packed for
    the gather must be those of the re-rendered frame.  One rank, forced overflow (OVR_HIP_POOL_CHUNKS=8)."""
    import torch
    case = make_case(ovr, oracle, n=32, tf="dense", cam="oblique", size=(96, 64), shading=2)
    ref_ren = hip_setup(ovr, hip_renderer_factory(), case)
    ref_ren.render()
    ref, _ = hip_frame(ovr, ref_ren)
    ref_ren.close()
    os.environ["OVR_HIP_POOL_CHUNKS"] = "8"
    try:
        ren = hip_renderer_factory()
        ren.set_image_shard(0, 1, 16, 16)
        hip_setup(ovr, ren, case)
        slots = ovr.tiles.max_owned_tiles(96, 64, 16, 16, 1)
        payload = torch.zeros((slots, 16, 16, 4), dtype=torch.float32, device="cuda")
        ren.render_async()
        ovr._lib.check(ren._lib.ovr_hip_pack_tiles(ren._h, C.c_void_p(payload.data_ptr()), payload.numel() * 4))
        ren.sync()
        assert ren.stats().pool_chunks > 8
    finally:
        del os.environ["OVR_HIP_POOL_CHUNKS"]
    frame = np.zeros((64, 96, 4), np.float32)
    ovr.tiles.unpack_tiles_host(payload.cpu().numpy(), frame, 16, 16, 0, 1)
    assert np.array_equal(frame.reshape(ref.shape), ref)
    ren.close()


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` (no launcher) starts two ranks itself; rehearsed on one card over gloo"""
    env = dict(os.environ, OVR_BENCH_BACKEND="gloo", OVR_BENCH_ONE_GPU="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    c, d, _ = run_bench(["--gpus", "2", "--config", "tiny", "--steps", "4", "--warmup", "1"], env=env, timeout=600)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"].startswith("gloo") and d["value"] > 0
    # the stdout line: the contract's fields and a per-rank summary; the tables are in the detail file
    assert c["n_gpus"] == 2 and c["rccl_ranks"] == 2 and c["steps"] == 4 and c["warmup"] == 1 and abs(c["value"] - d["value"]) <= 1e-4 * d["value"]
    assert set(c["ranks"]["step_ms"]) == {"min", "mean", "max"} and c["detail"].endswith(".json") and "roofline" in c
    assert d["per_frame"]["rays"] == 256 * 256
    # round 3: the N > 1 line carries what a first multi-GPU run needs to explain itself
    rk = d["ranks"]
    for c in ("march_ms", "shade_ms", "composite_ms", "pack_ms", "gather_ms", "gather_wait_ms", "step_ms"):
        assert set(rk[c]) == {"min", "mean", "max"} and rk[c]["max"] >= rk[c]["min"] >= 0.0, c
    assert len(rk["per_rank"]["samples"]) == 2 and sum(rk["per_rank"]["samples"]) == d["per_frame"]["samples"]
    assert rk["work_imbalance_max_over_mean"] >= 1.0 and rk["payload_bytes_per_rank"] > 0 and d["device_count"] >= 1
    # a launcher / --gpus mismatch is an error, not a silent single-rank run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "tiny"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stderr


def test_one_rank_of_n_as_a_profilable_stand_in():
    """`bench.py --shard-of N`: one process renders rank 0's image shard of N ranks without a gather - what tools/r03_prof_all.sh profiles so
    that the N-GPU line of the driver's run can quote counters (filed under world = N in profiles/*_traffic.json)"""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    res = {}
    for extra in ([], ["--shard-of", "2"], ["--shard-of", "2", "--shard-rank", "1"]):
        _, res[tuple(extra)], _ = run_bench(["--config", "tiny", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-views", "--no-skip-leg"] + extra, env=env, timeout=600)
    whole, r0, r1 = res[()], res[("--shard-of", "2")], res[("--shard-of", "2", "--shard-rank", "1")]
    assert "stand-in" in r0["config"]["parallelism"] and "stand-in" not in whole["config"]["parallelism"] and r0["n_gpus"] == 1
    assert r0["per_frame"]["samples"] + r1["per_frame"]["samples"] == whole["per_frame"]["samples"]
    assert r0["per_frame"]["rays"] + r1["per_frame"]["rays"] == whole["per_frame"]["rays"]
    src = r0["roofline"]["traffic_source"]
    assert "|2|2|" in src or "shard of 2" in src, src   # looked up under world = 2, never under the whole frame's key


@pytest.mark.parametrize("config,n", [("c4", 96), ("c5", 64)])
def test_two_ranks_on_one_card_at_the_8_gpu_configurations_shapes(config, n):
    """the N > 1 path of BASELINE's 8-GPU configurations (C4: u16, 1080p; C5: 4K, blue-noise jitter, accumulation) at a reduced volume
    edge, two gloo ranks on one card: the frame's work is the sum of the ranks' and the line carries the per-rank report"""
    env = dict(os.environ, OVR_BENCH_BACKEND="gloo", OVR_BENCH_ONE_GPU="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    _, d, _ = run_bench(["--gpus", "2", "--config", config, "--n", str(n), "--steps", "3", "--warmup", "1"], env=env)
    w, h = (1920, 1080) if config == "c4" else (3840, 2160)
    assert d["n_gpus"] == 2 and d["per_frame"]["rays"] == w * h and d["dtype"] == ("u16" if config == "c4" else "f32")
    assert len(d["ranks"]["per_rank"]["kernel_ms"]) == 2 and d["ranks"]["work_imbalance_max_over_mean"] < 1.5
    if config == "c5":
        # progressive accumulation gathered only when it is mapped (SURVEY 8e on C5): same frames, one gather at the end of the timed region
        _, d2, _ = run_bench(["--gpus", "2", "--config", config, "--n", str(n), "--steps", "3", "--warmup", "1", "--gather-every", "64"], env=env)
        assert d2["gather"].startswith("every 64th") and d["gather"] == "every frame" and d2["per_frame"] == d["per_frame"]


# ---- view-dependent volume replicas (thin layouts) ----------------------------------------------------------------------------

_AXIS_CAMS = {"x": ((200.0, 3.0, -2.0), (0.0, 1.0, 0.0)), "y": ((4.0, 190.0, 5.0), (0.0, 0.0, 1.0)), "z": ((-3.0, 2.0, 210.0), (0.0, 1.0, 0.0)),
              "oblique": ((-120.0, 70.0, 66.0), (0.0, 1.0, 0.0))}


def _layout_case(ovr, oracle, dtype, dims, cam, shading=2, size=(88, 56)):
    case = make_case(ovr, oracle, dtype=dtype, dims=dims, tf="bumps", size=size, shading=shading)
    c = np.array(dims, dtype=np.float64) / 2.0
    off, up = _AXIS_CAMS[cam]
    case["cam"] = (tuple(c + np.array(off)), tuple(c), up)
    return case


@pytest.mark.parametrize("dtype", [np.float32, np.uint16, np.float64, np.uint8])
@pytest.mark.parametrize("cam", ["x", "y", "z", "oblique"])
def test_layouts_are_bit_identical(ovr, oracle, hip_renderer_factory, dtype, cam):
    """the general layout, the two thin replicas (pair axis x / pair axis y; float and 16-bit volumes) and the quad replica (round 3: a tap is
    two loads; float, 16-bit and 8-bit volumes) of a non-cubic volume give the same frame bit for bit - with and without empty-space skipping, both pipelines -
    and that frame agrees with the oracle"""
    dims = (45, 70, 33)   # nx != ny != nz: an exchanged axis cannot go unnoticed
    case = _layout_case(ovr, oracle, dtype, dims, cam)
    ren = hip_renderer_factory()
    ren.set_volume_layouts(2)
    hip_setup(ovr, ren, case)
    frames = {}
    for choice in (0, 1, 2, 3) if np.dtype(dtype) != np.uint8 else (0, 3):
        for skip, pipeline in ((False, 0), (True, 0), (False, 1)):
            ren.set_layout_choice(choice)
            ren.set_empty_space_skipping(skip)
            ren.set_shading_pipeline(pipeline)
            ren.commit()
            ren.render()
            st = ren.stats()
            assert st.layout == choice
            frames[(choice, skip, pipeline)] = (hip_frame(ovr, ren), (st.samples + st.skipped_samples, st.shaded_samples, st.shadow_samples + st.skipped_shadow_samples))
    (ref_rgba, ref_grad), ref_cnt = frames[(0, False, 0)]
    for k, ((rgba, grad), cnt) in frames.items():
        assert np.array_equal(rgba, ref_rgba) and np.array_equal(grad, ref_grad), k
        assert cnt == ref_cnt, k
    o_rgba, _, cnt = oracle_scene(oracle, case).render()
    assert ref_cnt[0] == cnt.samples and ref_cnt[1] == cnt.shaded_samples and cnt.shaded_samples > 100
    compare(oracle, ref_rgba, o_rgba, name=f"layouts {np.dtype(dtype).name} {cam}")
    info = ren.volume_info()
    # general (x 4/3) + two thin (x 2 each, not for 8-bit) + quad (x 4)
    assert info.resident_bytes > (5 if np.dtype(dtype) == np.uint8 else 9) * case["vol"].size * min(case["vol"].itemsize, 4)
    ren.close()


def _render_until_layout(ovr, ren, layout, limit=500):
    """replicas are built in the background (round 4): frames read the general layout until the one the rule asks for is resident -
    every frame on the way is the same frame bit for bit"""
    import time
    ren.render()
    first = hip_frame(ovr, ren)
    seen = [ren.stats().layout]
    for _ in range(limit):
        st = ren.stats()
        if st.layout == layout and st.replicas_building == 0:
            break
        time.sleep(0.002)
        ren.render()
        seen.append(ren.stats().layout)
        rgba, grad = hip_frame(ovr, ren)
        assert np.array_equal(rgba, first[0]) and np.array_equal(grad, first[1])
    return seen


def test_layout_follows_the_camera(ovr, oracle, hip_renderer_factory):
    """within ~18 degrees of a volume axis the frame reads a thin replica (pair axis off that axis) - once it has been built in the background;
    types without replicas and renderers told not to build them stay on the general layout"""
    dims = (40, 40, 40)
    expect = {"x": 2, "y": 1, "z": 2, "oblique": 0}   # along z either thin replica serves: the one wide in the drift direction (x here)
    for cam, layout in expect.items():
        case = _layout_case(ovr, oracle, np.float32, dims, cam, shading=1, size=(48, 32))
        ren = hip_setup(ovr, hip_renderer_factory(), case)
        seen = _render_until_layout(ovr, ren, layout)
        assert ren.stats().layout == layout and set(seen) <= {0, layout}, (cam, seen)
        ren.close()
    case = _layout_case(ovr, oracle, np.uint8, dims, "z", shading=1, size=(48, 32))
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    ren.render()
    assert ren.stats().layout == 0
    ren.close()
    case = _layout_case(ovr, oracle, np.float32, dims, "z", shading=1, size=(48, 32))
    ren = hip_renderer_factory()
    ren.set_volume_layouts(0)
    hip_setup(ovr, ren, case)
    general_only = ren.volume_info().resident_bytes
    ren.render()
    assert ren.stats().layout == 0
    ren.close()
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    assert ren.volume_info().resident_bytes == general_only   # (round 4) nothing but the general layout until a frame asks for a replica
    _render_until_layout(ovr, ren, 2)
    assert ren.stats().layout == 2 and 2 * general_only < ren.volume_info().resident_bytes < 3 * general_only   # + the one thin replica
    ren.close()
    ren = hip_renderer_factory()
    ren.set_volume_layouts(2)                                    # mode 2: every replica resident when ovr_hip_set_volume returns
    hip_setup(ovr, ren, case)
    assert ren.volume_info().resident_bytes > 6 * general_only
    ren.render()
    assert ren.stats().layout == 2 and ren.stats().replicas_building == 0
    ren.close()


@pytest.mark.parametrize("dtype", [np.float32, np.uint16, np.uint8])
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_quad_layout_in_every_addressing_mode(ovr, oracle, hip_renderer_factory, mode, dtype):
    """the quad replica under the 32-bit element, 64-bit z-table and computed addressing modes; in place and pooled, with skipping"""
    case = _layout_case(ovr, oracle, dtype, (37, 29, 50), "oblique", shading=2, size=(64, 40))
    ren = hip_renderer_factory()
    ren.set_volume_layouts(2)
    hip_setup(ovr, ren, case)
    ren.set_layout_choice(0)
    ren.commit()
    ren.render()
    ref = hip_frame(ovr, ren)[0]
    os.environ["OVR_HIP_ADDRESSING"] = str(mode)
    try:
        for skip, pipeline in ((False, 1), (False, 2), (True, 0)):
            ren.set_layout_choice(3)
            ren.set_empty_space_skipping(skip)
            ren.set_shading_pipeline(pipeline)
            ren.commit()
            ren.render()
            assert ren.stats().layout == 3
            assert np.array_equal(hip_frame(ovr, ren)[0], ref), (mode, skip, pipeline)
    finally:
        del os.environ["OVR_HIP_ADDRESSING"]
    ren.close()


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_thin_layouts_in_every_addressing_mode(ovr, oracle, hip_renderer_factory, mode):
    """the 32-bit element, 64-bit z-table and computed addressing modes forced onto the thin replicas of a small volume"""
    case = _layout_case(ovr, oracle, np.uint16, (37, 29, 50), "y", shading=2, size=(64, 40))
    ren = hip_renderer_factory()
    ren.set_volume_layouts(2)
    hip_setup(ovr, ren, case)
    ren.render()
    ref = hip_frame(ovr, ren)[0]
    os.environ["OVR_HIP_ADDRESSING"] = str(mode)
    try:
        for choice in (1, 2):
            ren.set_layout_choice(choice)
            ren.commit()
            ren.render()
            assert ren.stats().layout == choice
            assert np.array_equal(hip_frame(ovr, ren)[0], ref), (mode, choice)
    finally:
        del os.environ["OVR_HIP_ADDRESSING"]
    ren.close()


# ---- LDS-staged bricks (north_star) ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("cam,spp,size,n", [("front", 1, (64, 64), 48), ("oblique", 1, (96, 56), 48), ("inside", 2, (72, 40), 40), ("oblique", 3, (333, 77), 64),
                                            ("top", 1, (128, 128), 96)])
def test_lds_staged_bricks_are_bit_identical(ovr, oracle, hip_renderer_factory, cam, spp, size, n):
    """the unshaded march with its bricks staged through LDS gives the frame of the ordinary march bit for bit (taps outside the
    staged box and rounds that do not fit take the ordinary path), and that frame agrees with the oracle"""
    case = make_case(ovr, oracle, n=n, tf="bumps", cam=cam, size=size, shading=0, spp=spp)
    ren = hip_renderer_factory()
    ren.set_layout_choice(0)   # the staged variant exists for the general layout (an axis view would otherwise read a thin replica - and stage nothing)
    hip_setup(ovr, ren, case, accumulate=True)
    ren.render(); ren.render()
    plain, plain_g = hip_frame(ovr, ren)
    st0 = ren.stats()
    ren.set_lds_staging(True)
    ren.set_camera(ovr.Camera(*case["cam"], case["fovy"]))   # reset the accumulation
    ren.commit()
    ren.render(); ren.render()
    staged, staged_g = hip_frame(ovr, ren)
    st1 = ren.stats()
    assert np.array_equal(plain, staged) and np.array_equal(plain_g, staged_g)
    assert (st0.samples, st0.shaded_samples, st0.rays) == (st1.samples, st1.shaded_samples, st1.rays)
    assert st1.skipped_samples == 0 and st1.lds_fallback_taps == 0      # no tap of the STAGED run fell outside its box
    assert st0.lds_rounds == 0 and st1.lds_rounds > 0                    # ... and the staged variant did run
    if cam in ("front", "top"):                                          # rays along an axis: the box of a round fits, bricks ARE staged
        assert st1.lds_rounds > st1.lds_unstaged_rounds, (st1.lds_rounds, st1.lds_unstaged_rounds)
    ref, _, cnt = oracle_scene(oracle, case).render(frames=2, accumulate=True)
    assert st1.samples == cnt.samples
    compare(oracle, staged, ref, name=f"lds staging {cam}")
    # shaded modes and skipping do not have the variant: the switch is ignored there
    ren.set_shading(2)
    ren.commit()
    ren.render()
    assert ren.stats().lds_fallback_taps == 0 and ren.stats().lds_unstaged_rounds == 0
    ren.close()


def test_skipping_kernels_are_suspended_where_nothing_is_skipped(ovr, oracle, hip_renderer_factory):
    """With empty-space skipping enabled a frame that skipped < 10 % of its sample steps (here: a dense transfer function - every
    macrocell can hold opacity) switches the renderer to the plain kernels; a transfer-function change probes the skipping kernels at
    once, and a scene with empty space keeps them.  Frames are bit-identical whichever kernels run."""
    case = make_case(ovr, oracle, n=48, tf="dense", cam="oblique", size=(96, 72), shading=2)
    plain = hip_setup(ovr, hip_renderer_factory(), case, accumulate=True)
    for _ in range(4):
        plain.render()
    want = hip_frame(ovr, plain)[0]
    ren = hip_setup(ovr, hip_renderer_factory(), case, accumulate=True)
    ren.set_empty_space_skipping(True)
    ren.commit()
    used = []
    for _ in range(4):
        ren.render()
        st = ren.stats()
        used.append(st.skipping_kernels)
        assert st.samples + st.skipped_samples == plain.stats().samples
    assert used == [1, 0, 0, 0], used
    assert np.array_equal(hip_frame(ovr, ren)[0], want)          # 4 accumulated frames each
    colors, alphas, vr = ovr.synth.make_tfn("sparse", 1024, np.float32)
    ren.set_transfer_function(colors, alphas, vr)                # what is empty has changed: probe at once - and keep them
    ren.commit()
    used = []
    for _ in range(3):
        ren.render()
        used.append(ren.stats().skipping_kernels)
    assert used == [1, 1, 1] and ren.stats().skipped_samples > 0, used
    sparse_case = dict(case, colors=colors, alphas=alphas, vr=vr)
    ref = hip_setup(ovr, hip_renderer_factory(), sparse_case, accumulate=True)
    for _ in range(3):
        ref.render()
    assert np.array_equal(hip_frame(ovr, ren)[0], hip_frame(ovr, ref)[0])
