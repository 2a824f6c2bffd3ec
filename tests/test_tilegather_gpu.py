"""The N > 1 frame on one card: two gloo ranks (both on cuda:0) render their image tiles with the HIP kernels and run the
pipelined TileGather (pack -> gather on its own stream -> deferred scatter) over several accumulated frames; rank 0's frame
must equal the unsharded frame bit for bit.  gloo stands in for RCCL (two ranks cannot share one GPU under RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, SIZE, TILE, FRAMES = 96, (200, 120), 16, 5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _renderer(ovr, shard=None):
    import torch
    vol = ovr.synth.make_volume_torch(N, torch.device("cuda", 0), "float32")
    colors, alphas, vr = ovr.synth.make_tfn("bumps", 256)
    eye, at, up = ovr.synth.make_camera("oblique", N)
    ren = ovr.create_renderer("hip")
    ren.set_fbsize(SIZE)
    ren.set_frame_accumulation(True)
    ren.set_sample_per_pixel(2)
    ren.set_shading(2)
    ren.set_transfer_function(colors, alphas, vr)
    if shard:
        ren.set_image_shard(*shard)
    ren.init(ovr.Scene(volume=vol, transfer_function=None), ovr.Camera(eye, at, up))
    ren.commit()
    return ren


def _worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import ovr_amd as ovr
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        ren = _renderer(ovr, shard=(rank, world, TILE, TILE))
        g = ovr.tiles.TileGather(ren, SIZE[0], SIZE[1], TILE, rank, world, torch.device("cuda", 0))
        snaps = []
        for i in range(FRAMES):
            ren.render_async()
            g.run()
            ren.sync()
            if rank == 0 and i == FRAMES - 2:
                snaps.append(g.frame.clone())   # run() number i scatters frame i - 1: this snapshot must be frame FRAMES - 3
                # (the clone is a kernel on torch's default stream, which the gatherer's non-blocking streams do not wait for: without this
                # synchronisation the NEXT run()'s scatter could overtake it on a busy card - seen twice in ~20 full-suite runs, always with 4 ranks)
                torch.cuda.synchronize()
        g.flush()
        ren.sync()
        torch.cuda.synchronize()
        if rank == 0:
            np.save(out_path, g.frame.cpu().numpy())
            np.save(out_path + ".prev.npy", snaps[0].cpu().numpy())
        ren.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])   # + this process: at most 5 processes on the card
def test_pipelined_gather_rebuilds_the_accumulated_frame(tmp_path, ovr, world):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got, prev = np.load(out), np.load(out + ".prev.npy")
    ren = _renderer(ovr)
    fb = ovr.FrameBufferData()
    frames = []
    for _ in range(FRAMES):
        ren.render()
        ren.mapframe(fb)
        frames.append(np.array(fb.rgba.data(), copy=True).reshape(SIZE[1], SIZE[0], 4))
    ren.close()
    def which(frame):
        """per 16x16 tile: the index of the accumulated frame its pixels equal (-1: none) - what a failure prints"""
        out = {}
        for ty in range(0, SIZE[1], TILE):
            for tx in range(0, SIZE[0], TILE):
                sl = (slice(ty, ty + TILE), slice(tx, tx + TILE))
                m = [k for k in range(FRAMES) if np.array_equal(frame[sl], frames[k][sl])]
                key = (m[-1] if m else -1, ((tx // TILE) + (ty // TILE)) % world)
                out[key] = out.get(key, 0) + 1
        return out   # {(frame index, owning rank): tiles}
    assert np.array_equal(got, frames[-1]), which(got)
    # after run() number FRAMES - 1 (index FRAMES - 2) and the renderer's sync, the scattered frame is the one before it
    assert np.array_equal(prev, frames[FRAMES - 3]), which(prev)
