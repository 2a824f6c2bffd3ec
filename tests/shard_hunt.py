"""One-off hunt: image shards of random frame sizes, tile shapes and world sizes (1 ... 8) - every rank's frame rendered on one card, packed, and
scattered into one frame, which must equal the unsharded frame bit for bit; the payload of each rank must equal the host restatement
(tiles.pack_tiles_host).   usage: python tests/shard_hunt.py [cases] [seed]"""
import ctypes as C
import sys
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + '/tests', _R + '/oracle']
import numpy as np
import torch
import ovr_amd as ovr
import oracle as O
from helpers import make_case, hip_setup, hip_frame

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for i in range(cases):
    size = (int(rng.integers(1, 200)), int(rng.integers(1, 130)))
    world = int(rng.integers(1, 9))
    tw, th = int(rng.choice([1, 3, 8, 16, 17, 32, 64])), int(rng.choice([1, 2, 8, 16, 24, 64]))
    case = make_case(ovr, O, n=20, tf=str(rng.choice(["bumps", "dense"])), cam=str(rng.choice(["oblique", "inside", "front"])), size=size, shading=int(rng.integers(0, 3)))
    full = hip_setup(ovr, ovr.create_renderer("hip"), case)
    full.render()
    ref = hip_frame(ovr, full)[0]
    slots = ovr.tiles.max_owned_tiles(size[0], size[1], tw, th, world)
    frame = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda")
    ok = True
    root = ovr.create_renderer("hip")             # the rank that scatters: same world and tile shape as the senders
    root.set_image_shard(0, world, tw, th)
    hip_setup(ovr, root, case)
    for rank in range(world):
        ren = ovr.create_renderer("hip")
        ren.set_image_shard(rank, world, tw, th)
        hip_setup(ovr, ren, case, pipeline=int(rng.integers(0, 3)))
        ren.render()
        payload = torch.zeros((max(slots, 1), th, tw, 4), dtype=torch.float32, device="cuda")
        ovr._lib.check(ren._lib.ovr_hip_pack_tiles(ren._h, C.c_void_p(payload.data_ptr()), payload.numel() * 4))
        ren.sync(); torch.cuda.synchronize()
        host = ovr.tiles.pack_tiles_host(ref, tw, th, rank, world, slots=max(slots, 1))
        if not np.array_equal(payload.cpu().numpy(), host):
            ok = False
        ovr._lib.check(root._lib.ovr_hip_unpack_tiles(root._h, rank, C.c_void_p(payload.data_ptr()), payload.numel() * 4, C.c_void_p(frame.data_ptr()), frame.numel() * 4))
        root.sync(); torch.cuda.synchronize()
        ren.close()
    if not (ok and np.array_equal(frame.cpu().numpy(), ref)):
        bad += 1
        print(f"case {i}: size {size} world {world} tile {tw}x{th}: payload ok {ok}, reassembled == unsharded {np.array_equal(frame.cpu().numpy(), ref)}", flush=True)
    full.close(); root.close()
print(f"{cases} shard configurations, {bad} differ")
