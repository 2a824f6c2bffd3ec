"""Parity at BASELINE.json's full sizes through size-independent properties (the CPU oracle would need minutes per frame there):
pipelines and options that must not change a single bit, conservation of the sample counters, sharded == unsharded, and the
oracle itself on a 1/64 subset of the pixels of the full-size frame."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(ovr, ren, vol, n, size, shading, tf="sparse", cam="oblique", accumulate=False, pipeline=0, skip=False, shard=None, dtype=np.float32):
    colors, alphas, vr = ovr.synth.make_tfn(tf, 1024, dtype)
    eye, at, up = ovr.synth.make_camera(cam, n)
    ren.set_fbsize(size)
    ren.set_frame_accumulation(accumulate)
    ren.set_shading(shading)
    ren.set_shading_pipeline(pipeline)
    ren.set_empty_space_skipping(skip)
    ren.set_transfer_function(colors, alphas, vr)
    if shard:
        ren.set_image_shard(*shard)
    ren.init(ovr.Scene(volume=vol, transfer_function=None), ovr.Camera(eye, at, up))
    ren.commit()
    return ren


def _frame(ovr, ren):
    fb = ovr.FrameBufferData()
    ren.mapframe(fb)
    return np.array(fb.rgba.data(), copy=True)


@pytest.fixture(scope="module")
def c3_volume(ovr):
    import torch
    return ovr.synth.make_volume_torch(1024, torch.device("cuda", 0), "float32")


def test_c3_invariants(ovr, oracle, hip_renderer_factory, c3_volume):
    """C3: 1024^3 f32, 1920x1080, reference shading.  pooled == in place == pooled + skipping, bit for bit; counters conserved;
    accumulating identical frames leaves the frame unchanged; frame is finite with alpha in [0, 1]"""
    n, size = 1024, (1920, 1080)
    ref = stats_ref = None
    for pipeline, skip in ((2, False), (1, False), (2, True)):
        ren = _setup(ovr, hip_renderer_factory(), c3_volume, n, size, 2, pipeline=pipeline, skip=skip, accumulate=True)
        ren.render()
        f, st = _frame(ovr, ren), ren.stats()
        if ref is None:
            ref, stats_ref = f, st
            assert np.isfinite(f).all() and f[..., 3].min() >= 0.0 and f[..., 3].max() <= 1.0
            assert st.rays == size[0] * size[1] and st.samples > 2e8 and st.shadow_samples > 1e8
            ren.render()
            ren.render()
            assert ren.stats().frame_index == 3
            f3 = _frame(ovr, ren)
            assert np.abs(f3 - f).max() <= 2e-7      # (x + x + x) / 3
        else:
            assert np.array_equal(f, ref), (pipeline, skip)
            assert st.samples + st.skipped_samples == stats_ref.samples
            assert st.shaded_samples == stats_ref.shaded_samples
            assert st.shadow_samples + st.skipped_shadow_samples == stats_ref.shadow_samples
        ren.close()


def test_c3_shade_heavy_regime_at_full_size(ovr, oracle, hip_renderer_factory, c3_volume):
    """C3's volume under a dense transfer function at the scene files' sampling rate 4 (round 3: every sample shaded, long shadow
    marches): general layout in place == quad replica pooled == whatever the renderer's measuring settles on, bit for bit, with the same
    counters; the measured choice is not slower than the rules' (it timed them); and the oracle agrees on 1/64 of the tiles"""
    n, size, tile = 1024, (1920, 1080), 64
    frames = {}
    for tag, layout, pipeline, tune in (("general in place", 0, 1, "0"), ("quad pooled", 3, 2, "0"), ("measured", -1, 0, "1")):
        os.environ["OVR_HIP_TUNE"] = tune
        try:
            ren = hip_renderer_factory()
        finally:
            del os.environ["OVR_HIP_TUNE"]
        ren.set_layout_choice(layout)
        ren.set_volume_sampling_rate(4.0)
        _setup(ovr, ren, c3_volume, n, size, 2, tf="dense", cam="front", pipeline=pipeline)
        ren.set_volume_sampling_rate(4.0)
        ren.commit()
        ms = []
        for _ in range(12 if tag == "measured" else 2):
            ren.render()
            ms.append(ren.stats().kernel_ms)
        # (round 4) the quad replica the tuner wants is allocated and built in the background - a fresh 17.6 GB hipMalloc can take most of a second -
        # and nothing is probed meanwhile: render on until the measurement has been made
        for _ in range(80):
            st = ren.stats()
            if tag != "measured" or (st.tuning == 2 and st.replicas_building == 0):
                break
            ren.render()
            ms.append(ren.stats().kernel_ms)
        st = ren.stats()
        frames[tag] = (_frame(ovr, ren), (st.samples, st.shaded_samples, st.shadow_samples), ms[-1], (st.layout, st.pipeline, st.tuning))
        ren.close()
    ref, cnt, ms_rules, _ = frames["general in place"]
    assert np.isfinite(ref).all() and cnt[1] == cnt[0] and cnt[2] > 50 * cnt[0]          # all shaded, > 50 shadow iterations per sample
    for tag, (f, c, _, _) in frames.items():
        assert np.array_equal(f, ref) and c == cnt, tag
    assert frames["quad pooled"][3][:2] == (3, 2) and frames["measured"][3][2] == 2
    assert frames["measured"][2] <= 1.05 * min(ms_rules, frames["quad pooled"][2]), {k: v[2:] for k, v in frames.items()}
    vol_host = c3_volume.cpu().numpy()
    colors, alphas, vr = ovr.synth.make_tfn("dense", 1024)
    sc = oracle.OracleScene(vol_host, colors, alphas, vr, ovr.synth.make_camera("front", n), size[0], size[1], rate=4.0, shading=oracle.SHADE_FULL,
                            shard=(23, 64, tile, tile), skip_zero_opacity=True)
    o_rgba, _, ocnt = sc.render()
    mask = np.zeros((size[1], size[0]), bool)
    for tx, ty in ovr.tiles.owned_tiles(size[0], size[1], tile, tile, 23, 64):
        mask[ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile] = True
    assert ocnt.samples > 1e5
    d8 = np.abs(oracle.rgba8(ref, flip=False).astype(int) - oracle.rgba8(o_rgba, flip=False).astype(int))[mask]
    assert d8.max() <= 1 and np.abs(ref - o_rgba)[mask].max() <= 2e-4


def test_c3_shards_and_oracle_subset(ovr, oracle, hip_renderer_factory, c3_volume):
    """C3 sharded over 8 'ranks' (run one after the other on this card) reassembles to the unsharded frame bit for bit; and the
    CPU oracle agrees on a subset of the full-size frame's pixels (the 17 64x64 tiles of one anti-diagonal, gradient shading)"""
    import torch
    n, size, tile = 1024, (1920, 1080), 64
    full = _setup(ovr, hip_renderer_factory(), c3_volume, n, size, 1)
    full.render()
    ref = _frame(ovr, full)
    world = 8
    slots = ovr.tiles.max_owned_tiles(size[0], size[1], tile, tile, world)
    frame = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda")
    total = 0
    root = None  # the gathering rank's renderer (it knows the shard geometry) scatters every rank's payload
    for rank in range(world):
        ren = _setup(ovr, hip_renderer_factory(), c3_volume, n, size, 1, shard=(rank, world, tile, tile))
        ren.render()
        total += ren.stats().samples
        payload = torch.zeros((slots, tile, tile, 4), dtype=torch.float32, device="cuda")
        ovr._lib.check(ren._lib.ovr_hip_pack_tiles(ren._h, C.c_void_p(payload.data_ptr()), payload.numel() * 4))
        ren.sync()
        root = root or ren
        ovr._lib.check(root._lib.ovr_hip_unpack_tiles(root._h, rank, C.c_void_p(payload.data_ptr()), payload.numel() * 4,
                                                      C.c_void_p(frame.data_ptr()), frame.numel() * 4))
        root.sync()
        if ren is not root:
            ren.close()
    torch.cuda.synchronize()
    got = frame.cpu().numpy()
    bad = np.argwhere(np.any(got != ref, axis=2))
    assert bad.size == 0, (len(bad), bad.min(axis=0), bad.max(axis=0), got[tuple(bad[0])], ref[tuple(bad[0])])
    assert total == full.stats().samples
    # oracle on 1/64 of the tiles of the same full-size frame (the anti-diagonal tx + ty == 23 through the image centre)
    vol_host = c3_volume.cpu().numpy()
    colors, alphas, vr = ovr.synth.make_tfn("sparse", 1024)
    cam = ovr.synth.make_camera("oblique", n)
    sc = oracle.OracleScene(vol_host, colors, alphas, vr, cam, size[0], size[1], shading=oracle.SHADE_GRADIENT, shard=(23, 64, tile, tile))
    o_rgba, _, cnt = sc.render()
    mask = np.zeros((size[1], size[0]), bool)
    for tx, ty in ovr.tiles.owned_tiles(size[0], size[1], tile, tile, 23, 64):
        mask[ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile] = True
    assert mask.sum() > 30000 and cnt.samples > 1e6
    d8 = np.abs(oracle.rgba8(ref, flip=False).astype(int) - oracle.rgba8(o_rgba, flip=False).astype(int))[mask]
    assert d8.max() <= 1
    assert np.abs(ref - o_rgba)[mask].max() <= 2e-4
    # (round 5) ... and the frame the headline benchmark TIMES - reference shading incl. the shadow march, the pooled pipeline the automatic
    # choice takes, with and without empty-space skipping - on the same anti-diagonal: the HIP renderer draws exactly that shard, so its counters are the
    # subset's and must equal the oracle's (primary and shaded exactly; the shadow march within the borderline class of tests/test_parity_exact_gpu.py)
    sc = oracle.OracleScene(vol_host, colors, alphas, vr, cam, size[0], size[1], shading=oracle.SHADE_FULL, shard=(23, 64, tile, tile), skip_zero_opacity=True)
    o_rgba, _, cnt = sc.render()
    for skip in (False, True):
        ren = _setup(ovr, hip_renderer_factory(), c3_volume, n, size, 2, shard=(23, 64, tile, tile), skip=skip)
        ren.render()
        got, st = _frame(ovr, ren), ren.stats()
        ren.close()
        assert st.pipeline == 2 and st.samples + st.skipped_samples == cnt.samples and st.shaded_samples == cnt.shaded_samples, (skip, st.samples, cnt.samples)
        assert cnt.shadow_samples_visible > 1e6 and abs(int(st.shadow_samples + st.skipped_shadow_samples) - int(cnt.shadow_samples_visible)) <= 4, (skip, st.shadow_samples, cnt.shadow_samples_visible)
        d8 = np.abs(oracle.rgba8(got, flip=False).astype(int) - oracle.rgba8(o_rgba, flip=False).astype(int))[mask]
        assert d8.max() <= 1 and np.abs(got - o_rgba)[mask].max() <= 2e-4, skip
        assert np.array_equal(got[mask], _headline_frame(ovr, hip_renderer_factory, c3_volume, n, size)[mask]), skip   # the shard's pixels are the unsharded headline frame's


_HEADLINE = {}


def _headline_frame(ovr, hip_renderer_factory, vol, n, size):
    """the unsharded frame of the headline configuration (rendered once per module)"""
    if "f" not in _HEADLINE:
        ren = _setup(ovr, hip_renderer_factory(), vol, n, size, 2)
        ren.render()
        _HEADLINE["f"] = _frame(ovr, ren)
        ren.close()
    return _HEADLINE["f"]


def test_c2_and_c4_sizes(ovr, oracle, hip_renderer_factory):
    """C2 (512^3 f32, 1024^2, no shading) and a C4-shaped case (u16, > 4 GiB of bricks: 64-bit-capable addressing modes)"""
    import torch
    vol = ovr.synth.make_volume_torch(512, torch.device("cuda", 0), "float32")
    a = _setup(ovr, hip_renderer_factory(), vol, 512, (1024, 1024), 0)
    a.render()
    b = _setup(ovr, hip_renderer_factory(), vol, 512, (1024, 1024), 0, skip=True)
    b.render()
    assert np.array_equal(_frame(ovr, a), _frame(ovr, b))
    assert a.stats().samples == b.stats().samples + b.stats().skipped_samples
    # (round 5) C2 as benchmarked against the oracle itself: the 16 64x64 tiles of the anti-diagonal tx + ty == 15 through the image centre
    tile = 64
    colors, alphas, vr = ovr.synth.make_tfn("sparse", 1024)
    sc = oracle.OracleScene(vol.cpu().numpy(), colors, alphas, vr, ovr.synth.make_camera("oblique", 512), 1024, 1024, shading=oracle.SHADE_NONE, shard=(15, 32, tile, tile))
    o_rgba, _, cnt = sc.render()
    c = _setup(ovr, hip_renderer_factory(), vol, 512, (1024, 1024), 0, shard=(15, 32, tile, tile))
    c.render()
    mask = np.zeros((1024, 1024), bool)
    for tx, ty in ovr.tiles.owned_tiles(1024, 1024, tile, tile, 15, 32):
        mask[ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile] = True
    got, whole = _frame(ovr, c), _frame(ovr, a)
    assert mask.sum() == 16 * tile * tile and cnt.samples > 1e6 and c.stats().samples == cnt.samples
    assert np.array_equal(got[mask], whole[mask])                       # the shard's pixels are the whole frame's
    assert np.abs(oracle.rgba8(got, flip=False).astype(int) - oracle.rgba8(o_rgba, flip=False).astype(int))[mask].max() <= 1
    assert np.abs(got - o_rgba)[mask].max() <= 2e-4
    del vol
    torch.cuda.empty_cache()
    # 1600 x 1280 x 1200 u16: 2.46 G voxels -> 6.6 GB bricked: element offsets still fit 32 bits (addressing mode 1) ...
    for dims in ((1600, 1280, 1200),):
        v = ovr.synth.make_volume_torch(max(dims), torch.device("cuda", 0), "uint16")[: dims[2], : dims[1], : dims[0]].contiguous()
        colors, alphas, vr = ovr.synth.make_tfn("sparse", 1024, np.uint16)
        frames = []
        for pipeline in (2, 1):
            ren = hip_renderer_factory()
            ren.set_fbsize((640, 360))
            ren.set_shading(2)
            ren.set_shading_pipeline(pipeline)
            ren.set_transfer_function(colors, alphas, vr)
            c = (dims[0] / 2, dims[1] / 2, dims[2] / 2)
            ren.init(ovr.Scene(volume=v, transfer_function=None), ovr.Camera((c[0] - 2500.0, c[1] + 1200.0, c[2] + 1300.0), c, (0, 1, 0)))
            ren.commit()
            ren.render()
            frames.append(_frame(ovr, ren))
            assert ren.stats().shaded_samples > 0
            ren.close()
        assert np.array_equal(frames[0], frames[1]) and np.isfinite(frames[0]).all()


def test_c5_size_spp_pipelines(ovr, oracle, hip_renderer_factory, c3_volume):
    """C5's frame size (3840x2160) with jittered samples per pixel and progressive accumulation: the pooled pipeline (one
    march/shade/composite pass per generation) equals the in-place pipeline bit for bit, counters included"""
    n, size = 1024, (3840, 2160)
    frames, stats = [], []
    for pipeline in (2, 1):
        ren = hip_renderer_factory()
        ren.set_sample_per_pixel(2)
        _setup(ovr, ren, c3_volume, n, size, 2, pipeline=pipeline, accumulate=True)
        ren.render()
        ren.render()
        frames.append(_frame(ovr, ren))
        stats.append(ren.stats())
        ren.close()
    assert np.array_equal(frames[0], frames[1])
    assert np.isfinite(frames[0]).all() and frames[0][..., 3].max() <= 1.0
    a, b = stats
    assert (a.rays, a.samples, a.shaded_samples, a.shadow_samples) == (b.rays, b.samples, b.shaded_samples, b.shadow_samples)
    assert a.rays == 2 * size[0] * size[1] and a.frame_index == 2


def test_c1_full_frame_vs_oracle(ovr, oracle, hip_renderer_factory):
    """C1's shape (256^3 uint8, 512x512 - renderbatch's default frame) in full: every pixel against the CPU oracle, reference
    shading incl. the shadow march; sample counters equal exactly"""
    import torch
    n, size = 256, (512, 512)
    vol = ovr.synth.make_volume_torch(n, torch.device("cuda", 0), "uint8")
    ren = _setup(ovr, hip_renderer_factory(), vol, n, size, 2, dtype=np.uint8)
    ren.render()
    got, st = _frame(ovr, ren), ren.stats()
    colors, alphas, vr = ovr.synth.make_tfn("sparse", 1024, np.uint8)
    cam = ovr.synth.make_camera("oblique", n)
    sc = oracle.OracleScene(vol.cpu().numpy(), colors, alphas, vr, cam, size[0], size[1], shading=oracle.SHADE_FULL)
    ref, _, cnt = sc.render()
    # the oracle marches a shadow ray for every sample like the reference; those of samples with opacity > 0 are the GPU's.
    # Primary counts are exact; a shadow ray may end one iteration apart when its alpha sits on the 0.9999 threshold: 1 of 5.67 M here.  Round 5 found
    # its cause: not the pow (all three restatements of __powf in the oracle count the same 5 670 167) but the 8-bit voxels being normalised behind the
    # filter instead of one by one - the exact-parity build, which normalises them like the texture unit, counts 5 670 167 too and gives this frame bit for
    # bit (tests/test_parity_exact_gpu.py).  The bar: what was measured, doubled
    assert (st.samples, st.shaded_samples) == (cnt.samples, cnt.shaded_samples)
    assert abs(int(st.shadow_samples) - int(cnt.shadow_samples_visible)) <= 2
    d8 = np.abs(oracle.rgba8(got, flip=False).astype(int) - oracle.rgba8(ref, flip=False).astype(int))
    assert d8.max() <= 1
    assert np.abs(got - ref).max() <= 2e-4


def test_64bit_addressing_vs_oracle(ovr, oracle, hip_renderer_factory):
    """a volume with more than 2^32 stored voxels (2048 x 2048 x 1100 uint8: 5.3 G with the x-apron) takes the 64-bit z-table
    addressing mode; the CPU oracle checks a subset of its tiles, both pipelines must agree bit for bit"""
    import torch
    dims = (2048, 2048, 1100)
    dev = torch.device("cuda", 0)
    z = torch.arange(dims[2], device=dev, dtype=torch.float32).view(-1, 1, 1)
    y = torch.arange(dims[1], device=dev, dtype=torch.float32).view(1, -1, 1)
    x = torch.arange(dims[0], device=dev, dtype=torch.float32).view(1, 1, -1)
    # smooth blobs + a high-frequency term that depends on all three coordinates (an addressing slip must show)
    vol = torch.empty(dims[::-1], dtype=torch.uint8, device=dev)
    for z0 in range(0, dims[2], 100):
        zz = z[z0:z0 + 100]
        r2 = ((x - 1024) / 900) ** 2 + ((y - 1000) / 800) ** 2 + ((zz - 550) / 500) ** 2
        v = 200.0 * torch.exp(-2.5 * r2) + 30.0 * torch.sin(0.37 * x + 0.23 * y + 0.41 * zz)
        vol[z0:z0 + 100] = v.clamp_(0, 255).to(torch.uint8)
        del r2, v
    colors, alphas, vr = ovr.synth.make_tfn("sparse", 1024, np.uint8)
    c = (dims[0] / 2, dims[1] / 2, dims[2] / 2)
    cam = ((c[0] - 2600.0, c[1] + 1500.0, c[2] + 1900.0), c, (0.0, 1.0, 0.0))
    size, tile = (640, 360), 40
    frames = []
    for pipeline in (2, 1):
        ren = hip_renderer_factory()
        ren.set_fbsize(size)
        ren.set_shading(2)
        ren.set_shading_pipeline(pipeline)
        ren.set_transfer_function(colors, alphas, vr)
        ren.init(ovr.Scene(volume=vol, transfer_function=None), ovr.Camera(*cam))
        ren.commit()
        ren.render()
        frames.append(_frame(ovr, ren))
        st = ren.stats()
        ren.close()
    assert np.array_equal(frames[0], frames[1]) and st.shaded_samples > 10000
    vol_host = vol.cpu().numpy()
    del vol
    torch.cuda.empty_cache()
    sc = oracle.OracleScene(vol_host, colors, alphas, vr, cam, size[0], size[1], shading=oracle.SHADE_FULL, shard=(0, 12, tile, tile))
    ref, _, cnt = sc.render()
    mask = np.zeros((size[1], size[0]), bool)
    for tx, ty in ovr.tiles.owned_tiles(size[0], size[1], tile, tile, 0, 12):
        mask[ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile] = True
    assert cnt.samples > 1e6 and cnt.shaded_samples > 1000
    d8 = np.abs(oracle.rgba8(frames[0], flip=False).astype(int) - oracle.rgba8(ref, flip=False).astype(int))[mask]
    assert d8.max() <= 1
    assert np.abs(frames[0] - ref)[mask].max() <= 2e-4


def test_c4_as_named(ovr, oracle, hip_renderer_factory):
    """BASELINE C4 as named: 2048^3 uint16 (native u16 in HBM: 11.4 G stored voxels -> the 64-bit z-table addressing mode),
    1920x1080, reference shading.  pooled == in place == pooled + skipping bit for bit, counters conserved, and the CPU oracle
    on the 1/64 of the frame's 64x64 tiles of one anti-diagonal (u16 is sampled as raw float: array.cpp:335-338)."""
    import torch
    n, size, tile = 2048, (1920, 1080), 64
    dev = torch.device("cuda", 0)
    vol = ovr.synth.make_volume_torch(n, dev, "uint16")
    ref = st_ref = None
    for pipeline, skip in ((2, False), (1, False), (2, True)):
        ren = _setup(ovr, hip_renderer_factory(), vol, n, size, 2, pipeline=pipeline, skip=skip, dtype=np.uint16)
        info = ren.volume_info()
        assert info.resident_bytes > 2 ** 34 and tuple(info.dims) == (n, n, n)   # > 16 GiB: u16 with the x apron
        ren.render()
        f, st = _frame(ovr, ren), ren.stats()
        if ref is None:
            ref, st_ref = f, st
            assert np.isfinite(f).all() and f[..., 3].min() >= 0.0 and f[..., 3].max() <= 1.0
            assert st.rays == size[0] * size[1] and st.samples > 4e8 and st.shaded_samples > 1e6 and st.shadow_samples > 1e8
        else:
            assert np.array_equal(f, ref), (pipeline, skip)
            assert st.samples + st.skipped_samples == st_ref.samples
            assert st.shaded_samples == st_ref.shaded_samples
            assert st.shadow_samples + st.skipped_shadow_samples == st_ref.shadow_samples
        ren.close()
    vol_host = vol.cpu().numpy()
    if vol_host.dtype != np.uint16:
        vol_host = vol_host.view(np.uint16)
    del vol
    torch.cuda.empty_cache()
    colors, alphas, vr = ovr.synth.make_tfn("sparse", 1024, np.uint16)
    cam = ovr.synth.make_camera("oblique", n)
    # the zero-opacity shortcut (bit-identical, tests/test_oracle_kat.py) keeps the oracle at seconds for ~2000-step rays
    sc = oracle.OracleScene(vol_host, colors, alphas, vr, cam, size[0], size[1], shading=oracle.SHADE_FULL, shard=(23, 64, tile, tile),
                            skip_zero_opacity=True)
    o_rgba, _, cnt = sc.render()
    mask = np.zeros((size[1], size[0]), bool)
    for tx, ty in ovr.tiles.owned_tiles(size[0], size[1], tile, tile, 23, 64):
        mask[ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile] = True
    assert mask.sum() > 30000 and cnt.samples > 1e7 and cnt.shaded_samples > 1e4
    d8 = np.abs(oracle.rgba8(ref, flip=False).astype(int) - oracle.rgba8(o_rgba, flip=False).astype(int))[mask]
    assert d8.max() <= 1
    assert np.abs(ref - o_rgba)[mask].max() <= 2e-4


def test_c5_as_named(ovr, oracle, hip_renderer_factory, c3_volume):
    """BASELINE C5 as named: 1024^3 f32, 3840x2160, 64 progressive frames of one blue-noise-jittered sample per pixel, accumulated.
    pooled == in place bit for bit after all 64 frames; the CPU oracle agrees on a tile subset after 2 accumulated frames."""
    n, size, tile = 1024, (3840, 2160), 64
    noise = ovr.synth.make_noise_tile(64)
    frames, stats, two = [], [], None
    for pipeline in (2, 1):
        ren = hip_renderer_factory()
        ren.set_noise_tile(noise)
        ren.set_pixel_jitter(ovr.JITTER_BLUE_NOISE)
        _setup(ovr, ren, c3_volume, n, size, 2, pipeline=pipeline, accumulate=True)
        for i in range(64):
            ren.render()
            if i == 1 and two is None:
                two = _frame(ovr, ren)
        frames.append(_frame(ovr, ren))
        stats.append(ren.stats())
        ren.close()
    a, b = stats
    assert a.frame_index == 64 and b.frame_index == 64
    assert np.array_equal(frames[0], frames[1])
    assert np.isfinite(frames[0]).all() and frames[0][..., 3].max() <= 1.0 + 1e-6
    assert (a.rays, a.samples, a.shaded_samples, a.shadow_samples) == (b.rays, b.samples, b.shaded_samples, b.shadow_samples)
    assert a.rays == size[0] * size[1]
    # the 64-frame image is smoother than a 2-frame one: jitter de-correlates over the slices
    assert not np.array_equal(two, frames[0])
    colors, alphas, vr = ovr.synth.make_tfn("sparse", 1024)
    cam = ovr.synth.make_camera("oblique", n)
    sc = oracle.OracleScene(c3_volume.cpu().numpy(), colors, alphas, vr, cam, size[0], size[1], shading=oracle.SHADE_FULL, jitter=1, noise=noise,
                            shard=(47, 128, tile, tile), skip_zero_opacity=True)
    o_rgba, _, cnt = sc.render(frames=2, accumulate=True)
    mask = np.zeros((size[1], size[0]), bool)
    for tx, ty in ovr.tiles.owned_tiles(size[0], size[1], tile, tile, 47, 128):
        mask[ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile] = True
    assert mask.sum() > 30000 and cnt.samples > 1e6
    d8 = np.abs(oracle.rgba8(two, flip=False).astype(int) - oracle.rgba8(o_rgba, flip=False).astype(int))[mask]
    assert d8.max() <= 1
    assert np.abs(two - o_rgba)[mask].max() <= 2e-4
