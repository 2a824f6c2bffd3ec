"""N > 1 path on CPU: two gloo ranks each render their image tiles (with the CPU oracle standing in for the GPU kernel),
pack them, gather to rank 0 through ovr_amd.tiles.gather_frame and rebuild the frame, which must equal the unsharded frame
bit for bit (pixels are independent and TEA seeds use the global pixel index)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path, spp):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import oracle as O
    import ovr_amd as ovr
    from helpers import make_case, oracle_scene
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W, H, TW, TH = 56, 40, 16, 8
        case = make_case(ovr, O, n=16, tf="bumps", cam="oblique", size=(W, H), shading=2, spp=spp)
        local, _, _ = oracle_scene(O, case, shard=(rank, world, TW, TH)).render(frames=2, accumulate=True, nthreads=1)
        slots = ovr.tiles.max_owned_tiles(W, H, TW, TH, world)
        payload = torch.from_numpy(ovr.tiles.pack_tiles_host(local, TW, TH, rank, world, slots))
        frame = np.zeros((H, W, 4), np.float32)

        def unpack(src, buf):
            ovr.tiles.unpack_tiles_host(buf.numpy(), frame, TW, TH, src, world)

        is_root = ovr.tiles.gather_frame(payload, W, H, TW, TH, rank, world, unpack)
        if is_root:
            np.save(out_path, frame)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("spp", [1, 2])
def test_two_rank_gather_rebuilds_the_frame(tmp_path, spp, ovr, oracle):
    from helpers import make_case, oracle_scene
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), out, spp), nprocs=2, join=True)
    got = np.load(out)
    case = make_case(ovr, oracle, n=16, tf="bumps", cam="oblique", size=(56, 40), shading=2, spp=spp)
    ref, _, _ = oracle_scene(oracle, case).render(frames=2, accumulate=True, nthreads=2)
    assert np.array_equal(got, ref)
