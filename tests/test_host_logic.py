"""host-side logic that needs no GPU: synthetic inputs, tile ownership, the scene flattening of the renderer mirror"""
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
