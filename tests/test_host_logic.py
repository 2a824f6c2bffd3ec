"""host-side logic that needs no GPU: synthetic inputs, tile ownership, the scene flattening of the renderer mirror"""
import os

import numpy as np
import pytest


def test_synthetic_volume_is_deterministic_and_typed(ovr):
    a = ovr.synth.make_volume(12)
    b = ovr.synth.make_volume(12)
    assert a.dtype == np.float32 and a.shape == (12, 12, 12) and np.array_equal(a, b)
    assert 0.0 <= a.min() and a.max() <= 1.0 and a.std() > 0.05
    u8 = ovr.synth.make_volume(12, np.uint8)
    assert np.array_equal(u8, np.rint(a * 255).astype(np.uint8))
    u16 = ovr.synth.make_volume(12, np.uint16)
    assert u16.dtype == np.uint16 and abs(float(u16.max()) / 65535 - float(a.max())) < 1e-4
    nz = ovr.synth.make_volume(0, dims=(10, 6, 4))
    assert nz.shape == (4, 6, 10)


def test_torch_and_numpy_generators_agree(ovr):
    import torch
    a = ovr.synth.make_volume(20)
    t = ovr.synth.make_volume_torch(20, torch.device("cpu")).numpy()
    assert np.abs(a - t).max() < 1e-6
    t8 = ovr.synth.make_volume_torch(20, torch.device("cpu"), "uint8").numpy()
    assert np.abs(t8.astype(int) - ovr.synth.make_volume(20, np.uint8).astype(int)).max() <= 1


def test_transfer_functions(ovr):
    for kind in ("sparse", "dense", "bumps"):
        c, a, vr = ovr.synth.make_tfn(kind, 64)
        assert c.shape == (64 * 3,) and a.shape == (64 * 2,) and vr == (0.0, 1.0)
        assert 0.0 <= a[1::2].min() and a[1::2].max() <= 1.0
        assert np.allclose(a[0::2], np.arange(64) / 63.0)
    assert ovr.synth.make_tfn("sparse", 16, np.uint8)[2] == (0.0, 255.0)
    assert ovr.synth.make_tfn("sparse", 16, np.uint16)[2] == (0.0, 65535.0)


def test_tile_ownership_matches_oracle(ovr, oracle):
    lib = oracle.load()
    for world in (1, 2, 3, 8):
        seen = set()
        for rank in range(world):
            for (tx, ty) in ovr.tiles.owned_tiles(200, 120, 32, 16, rank, world):
                assert lib.ovr_oracle_tile_owner(tx, ty, 7, world) == rank
                seen.add((tx, ty))
        assert len(seen) == 7 * 8  # every tile has exactly one owner


def test_pack_unpack_roundtrip(ovr):
    rng = np.random.default_rng(1)
    frame = rng.random((50, 70, 4), dtype=np.float32)
    out = np.zeros_like(frame)
    world = 3
    slots = ovr.tiles.max_owned_tiles(70, 50, 16, 16, world)
    for rank in range(world):
        payload = ovr.tiles.pack_tiles_host(frame, 16, 16, rank, world, slots)
        assert payload.shape == (slots, 16, 16, 4)
        ovr.tiles.unpack_tiles_host(payload, out, 16, 16, rank, world)
    assert np.array_equal(out, frame)


def test_scene_flattening_matches_reference_format(ovr):
    """MainRenderer::set_scene (reference ovr/renderer.h:299-341): colours -> flat RGB, opacities -> (i/(N-1), a) pairs"""
    captured = {}

    class Probe(ovr.DeviceHIP):
        def __init__(self):  # no library, no GPU: only the flattening logic is exercised
            self.current_scene = None

        def set_transfer_function(self, c, o, r):
            captured.update(c=np.asarray(c), o=np.asarray(o), r=tuple(r))

        def __del__(self):
            pass

    color = np.array([[0.1, 0.2, 0.3, 1.0], [0.4, 0.5, 0.6, 1.0], [0.7, 0.8, 0.9, 1.0]], np.float32)
    tf = ovr.TransferFunction(color=color, opacity=np.array([0.0, 0.5, 1.0], np.float32), value_range=(0.0, 255.0))
    Probe().set_scene(ovr.Scene(volume=None, transfer_function=tf))
    assert np.allclose(captured["c"], [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9])
    assert np.allclose(captured["o"], [0.0, 0.0, 0.5, 0.5, 1.0, 1.0])
    assert captured["r"] == (0.0, 255.0)


def test_camera_default_fovy_is_60(ovr):
    # renderer.h:149-152: set_camera(from, at, up) builds a Camera whose fovy is the default 60 (scene.h:219)
    assert ovr.Camera((0, 0, 1), (0, 0, 0), (0, 1, 0)).fovy == 60.0


def test_bench_refuses_a_launcher_mismatch_and_keys_traffic_by_the_kernel_hash(tmp_path, monkeypatch):
    """bench.py on the CPU: a WORLD_SIZE / --gpus mismatch is an error before anything touches the GPU, and a committed
    PMC traffic figure is only quoted for the kernel sources it was measured from"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="4", RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "tiny"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=4 but --gpus 2" in out.stderr
    sys.path.insert(0, root)
    import bench
    h = bench.kernels_hash()
    assert len(h) == 16 and h == bench.kernels_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    k3 = bench.traffic_key("c3", "oblique", "sparse", 2, 1, 1.0, 60.0, False)
    k2 = bench.traffic_key("c2", "oblique", "sparse", 0, 1, 1.0, 60.0, False)
    assert k3 == "c3|oblique|sparse|2|1|1.0|60.0|0"
    ctr = {"instantiation": "void ovrhip::raymarch_kernel<4, 2, 1, true, false, false, false>", "FETCH_SIZE": 1000.0, "WRITE_SIZE": 10.0, "GRBM_GUI_ACTIVE": 8 * 2.4e6,
           "SQ_INSTS_VMEM_RD": 2.0e7, "SQ_INSTS_VALU": 1.0e8, "TCP_TCC_READ_REQ_sum": 1.0e6, "TA_TA_BUSY_sum": 1.0e8, "mean_ms_rocprof": 1.0}
    entry = {"fetch_size_kib": {"raymarch_kernel": 1000}, "write_size_kib": {"raymarch_kernel": 10}, "traffic_bytes_per_launch": 2058240, "kernels": {"raymarch_kernel": ctr}}
    (prof / "r09_traffic.json").write_text(json.dumps({"kernels_hash": h, "entries": {k3: entry}}))
    (prof / "r08_traffic.json").write_text(json.dumps({"kernels_hash": "0" * 16, "entries": {k2: entry}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernels_hash", lambda: h)
    t, by_kernel, src = bench.load_traffic(k3)
    assert t == 2058240 and by_kernel["raymarch_kernel"]["FETCH_SIZE"] == 1000.0 and "r09_traffic.json" in src
    t, by_kernel, src = bench.load_traffic(k2)       # only measured for other kernels
    assert t is None and by_kernel == {} and src.startswith("null:")
    # the bound of a kernel from its counters: 2e7 gathers x 16 clk on 256 CUs over 2.4e6 clocks = 0.52 of the texture addressers, 1e8 vector
    # instructions x 4 clk on 1024 SIMDs = 0.16, 2 MB of traffic = nothing: bound by the gather rate - and never an HBM fraction above 1
    b = bench.kernel_bound(1.0, 5.0e9, ctr)
    assert b["bound"] == "ta" and abs(b["utilisation"]["ta"] - 2.0e7 * 16 / (256 * 2.4e6)) < 1e-3 and abs(b["frac"] - b["utilisation"]["ta"]) < 1e-3
    ctr_hbm = dict(ctr, FETCH_SIZE=3.0e6, SQ_INSTS_VMEM_RD=1.0e6)    # 6.3 GB of traffic in 1 ms for 5 GB of algorithmic bytes: the memory side
    assert bench.kernel_bound(1.0, 5.0e9, ctr_hbm)["bound"] == "hbm"
    assert bench.kernel_bound(1.0, 9.0e9, ctr_hbm)["bound"] != "hbm"   # more algorithmic bytes than traffic: the caches serve it, HBM is not the bound
    assert bench.kernel_bound(1.0, 5.0e9, None) is None
