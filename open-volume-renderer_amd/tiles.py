"""Image-plane sharding across one-process-per-GPU ranks.

Pixels are independent, so the path shards with no exchange during rendering: the image is cut into tiles, tile
(tx, ty) belongs to rank (tx + ty) % world, every rank renders its tiles against a full replica of the volume and
one gather per displayed frame brings the tiles to rank 0.  The only collective is that gather
(torch.distributed: RCCL over xGMI on GPUs, gloo on CPU for tests)."""
from typing import List, Optional

import numpy as np


def tile_owner(tx: int, ty: int, world: int) -> int:
    return (tx + ty) % world


def owned_tiles(width: int, height: int, tile_w: int, tile_h: int, rank: int, world: int) -> List[tuple]:
    """tiles of `rank` in the row-major order the pack kernel uses for its payload slots"""
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    return [(tx, ty) for ty in range(tiles_y) for tx in range(tiles_x) if tile_owner(tx, ty, world) == rank]


def max_owned_tiles(width, height, tile_w, tile_h, world) -> int:
    return max(len(owned_tiles(width, height, tile_w, tile_h, r, world)) for r in range(world))


def pack_tiles_host(frame: np.ndarray, tile_w: int, tile_h: int, rank: int, world: int, slots: Optional[int] = None) -> np.ndarray:
    """host restatement of the pack kernel: frame (H, W, 4) -> payload (slots, tile_h, tile_w, 4)"""
    h, w = frame.shape[:2]
    tiles = owned_tiles(w, h, tile_w, tile_h, rank, world)
    out = np.zeros((slots if slots is not None else len(tiles), tile_h, tile_w, 4), dtype=frame.dtype)
    for k, (tx, ty) in enumerate(tiles):
        blk = frame[ty * tile_h:(ty + 1) * tile_h, tx * tile_w:(tx + 1) * tile_w]
        out[k, :blk.shape[0], :blk.shape[1]] = blk
    return out


def unpack_tiles_host(payload: np.ndarray, frame: np.ndarray, tile_w: int, tile_h: int, rank: int, world: int) -> None:
    h, w = frame.shape[:2]
    for k, (tx, ty) in enumerate(owned_tiles(w, h, tile_w, tile_h, rank, world)):
        y0, x0 = ty * tile_h, tx * tile_w
        hh, ww = min(tile_h, h - y0), min(tile_w, w - x0)
        frame[y0:y0 + hh, x0:x0 + ww] = payload[k, :hh, :ww]


def gather_frame(local_payload, width, height, tile_w, tile_h, rank, world, unpack, dst: int = 0):
    """One gather to `dst` per displayed frame.  `local_payload` is this rank's (slots, th, tw, 4) tensor with the same
    `slots` on every rank (max over ranks); `unpack(src_rank, payload)` scatters one rank's payload into the final frame
    (ovr_hip_unpack_tiles on GPUs, unpack_tiles_host in the CPU tests).  Returns True on `dst`."""
    import torch
    import torch.distributed as dist
    if world == 1:
        unpack(0, local_payload)
        return True
    if rank == dst:
        bufs = [torch.empty_like(local_payload) for _ in range(world)]
        dist.gather(local_payload, gather_list=bufs, dst=dst)
        if local_payload.is_cuda:
            # RCCL runs on its own stream and torch only orders it against torch's current stream; the unpack kernels run on
            # the renderer's stream, so wait for the collective on the host before launching them
            torch.cuda.current_stream(local_payload.device).synchronize()
        for src in range(world):
            unpack(src, bufs[src])
        return True
    dist.gather(local_payload, gather_list=None, dst=dst)
    if local_payload.is_cuda:
        torch.cuda.current_stream(local_payload.device).synchronize()  # the payload buffer is reused by the next frame
    return False


class TileGather:
    """Per-frame gather of a sharded frame, pipelined over two HIP streams and with no host synchronisation of its own.

    The renderer is put on `render_stream`; the collective runs on `comm_stream`:

        render_stream:  render(i)  pack(i)              wait G(i-1)  unpack(i-1)   render(i+1)  pack(i+1)  wait G(i) ...
        comm_stream:               wait P(i)  gather(i) -> G(i)                                wait P(i+1)  gather(i+1) ...

    so the gather of frame i (RCCL over xGMI) overlaps the rendering of frame i+1 - payloads and receive buffers are
    double-buffered - and rank `dst` scatters frame i's tiles one call later.  Call `run()` after every
    `ren.render_async()`, `flush()` after the last frame; the caller's `ren.sync()` is the only host wait per frame.
    The reference keeps two framebuffer sets for the same reason (`swap`, device_impl.cpp:102-111)."""

    def __init__(self, ren, width, height, tile, rank, world, device, dst=0, time_every=1):
        import ctypes as C
        import torch
        self.C, self.torch = C, torch
        self.ren, self.rank, self.world, self.dst = ren, rank, world, dst
        self.render_stream = torch.cuda.Stream(device=device)
        self.comm_stream = torch.cuda.Stream(device=device)
        self.stream = self.render_stream
        ren.set_stream(self.render_stream.cuda_stream)
        slots = max_owned_tiles(width, height, tile, tile, world)
        root = rank == dst
        with torch.cuda.stream(self.render_stream):
            self.payload = [torch.zeros((slots, tile, tile, 4), dtype=torch.float32, device=device) for _ in range(2)]
            # receive side: one allocation per buffer set, rank k's payload at [k] - a single kernel scatters all of them
            self.recv = [torch.zeros((world,) + tuple(self.payload[0].shape), dtype=torch.float32, device=device) for _ in range(2)] if root else [None, None]
            self.bufs = [[r[k] for k in range(world)] for r in self.recv] if root else [None, None]
            self.frame = torch.zeros((height, width, 4), dtype=torch.float32, device=device) if root else None
        for t in self.payload + [r for r in self.recv if r is not None]:
            t.record_stream(self.comm_stream)   # allocated on one stream, used on both
        self.packed = [torch.cuda.Event() for _ in range(2)]
        self.gathered = [torch.cuda.Event() for _ in range(2)]
        # timing of the steps beside the rendering (round 3: the first multi-GPU run has to explain itself): event pairs per buffer set,
        # read back when the set is used again (two frames later - long finished, no host wait): pack, the collective as the
        # communication stream sees it (it includes waiting for the slowest peer), the render stream's stall on it, the scatter on rank dst
        ev = lambda: [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        self._t = {k: [ev(), ev()] for k in ("pack", "gather", "gather_wait", "unpack")}
        self._t_used = {k: [False, False] for k in self._t}
        self.times_ms = {k: 0.0 for k in self._t}   # sums over the frames whose events have been read
        self.times_n = {k: 0 for k in self._t}
        # an event record costs a few microseconds on both sides of the queue (profiles/r04_notes.md 15): with time_every = n only every n-th frame
        # carries the timing pairs (the means stay representative), the two events the pipeline itself needs are recorded on every frame
        self.time_every = max(1, int(time_every))
        self._timed = [True, True]
        self.payload_bytes = self.payload[0].numel() * 4
        self.index = 0
        self.outstanding = None   # buffer whose gather was enqueued and whose tiles are not scattered yet
        self.render_stream.synchronize()

    def _collect(self, key, b):
        """elapsed time of the event pair (key, b) of its previous use - two frames ago, complete unless the caller never synchronised"""
        if self._t_used[key][b]:
            e0, e1 = self._t[key][b]
            if e1.query():
                self.times_ms[key] += e0.elapsed_time(e1)
                self.times_n[key] += 1
            self._t_used[key][b] = False

    def collect_times(self):
        """after a device synchronisation: fold every finished event pair into times_ms; returns per-frame means in ms"""
        for key in self._t:
            for b in (0, 1):
                self._collect(key, b)
        return {k: (self.times_ms[k] / self.times_n[k] if self.times_n[k] else 0.0) for k in self._t}

    def _scatter_outstanding(self):
        from . import _lib as L
        C, ren, b = self.C, self.ren, self.outstanding
        if b is None:
            return
        timed = self._timed[b]
        if timed:
            self._collect("gather_wait", b)
            self._t["gather_wait"][b][0].record(self.render_stream)
        self.render_stream.wait_event(self.gathered[b])   # also frees payload[b] for the next pack into it
        if timed:
            self._t["gather_wait"][b][1].record(self.render_stream)
            self._t_used["gather_wait"][b] = True
        if self.rank == self.dst:
            with self.torch.cuda.stream(self.render_stream):
                recv = self.recv[b]
                if timed:
                    self._collect("unpack", b)
                    self._t["unpack"][b][0].record(self.render_stream)
                L.check(ren._lib.ovr_hip_unpack_all_tiles(ren._h, C.c_void_p(recv.data_ptr()), recv[0].numel() * 4, recv.numel() * 4,
                                                           C.c_void_p(self.frame.data_ptr()), self.frame.numel() * 4))
                if timed:
                    self._t["unpack"][b][1].record(self.render_stream)
                    self._t_used["unpack"][b] = True
        self.outstanding = None

    def run(self):
        """enqueue pack + gather of the frame just enqueued with ren.render_async(), and the scatter of the frame before it"""
        import torch.distributed as dist
        from . import _lib as L
        C, ren, torch = self.C, self.ren, self.torch
        b = self.index & 1
        timed = (self.index // 2) % self.time_every == 0   # (per buffer set: both sets of a timed pair of frames)
        if timed:
            self._collect("pack", b)
            self._collect("gather", b)
        with torch.cuda.stream(self.render_stream):
            if timed:
                self._t["pack"][b][0].record(self.render_stream)
            L.check(ren._lib.ovr_hip_pack_tiles(ren._h, C.c_void_p(self.payload[b].data_ptr()), self.payload[b].numel() * 4))
            if timed:
                self._t["pack"][b][1].record(self.render_stream)
                self._t_used["pack"][b] = True
            self.packed[b].record(self.render_stream)
        self._scatter_outstanding()
        self._timed[b] = timed
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(self.packed[b])
            if timed:
                self._t["gather"][b][0].record(self.comm_stream)
            dist.gather(self.payload[b], gather_list=self.bufs[b], dst=self.dst)
            if timed:
                self._t["gather"][b][1].record(self.comm_stream)
                self._t_used["gather"][b] = True
            self.gathered[b].record(self.comm_stream)
        self.outstanding = b
        self.index += 1

    def check(self):
        """after ren.sync(): a frame that overflowed its request pool although the pool had proven roomy was rendered again AFTER
        its tiles went into the gather - loud failure instead of a stale frame (the caller renders the frame again)"""
        if self.ren.stats().stale_tiles:
            raise RuntimeError("[hip] a frame was re-rendered after its tiles had been packed for the gather: gathered tiles are stale")

    def flush(self):
        """scatter the last frame's tiles (enqueued on the renderer's stream; ren.sync() or a device sync completes it)"""
        self._scatter_outstanding()
