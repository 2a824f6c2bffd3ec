"""Would overlapping the march of one part of a frame with the shading of another pay?  K renderers, each with rank k's image shard of K and its
own HIP stream, render C3 frames back to back (render_async: one frame in flight per stream, streams free-running against each other, so the
march of one shard meets the shade kernel of another); compared with ONE renderer of the whole frame.  A measurement, not a product path: the
reference's render() blocks per frame.   usage: python tools/overlap_probe.py [frames]"""
import sys
import time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + '/tests']
import torch, ovr_amd as ovr
from test_full_size_gpu import _setup

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n, size = 1024, (1920, 1080)
dev = torch.device('cuda', 0)
vol = ovr.synth.make_volume_torch(n, dev, 'float32')


def run(K, offset_ms=0.0):
    rens, streams = [], []
    for k in range(K):
        ren = ovr.create_renderer('hip')
        ren.set_volume_layouts(0)   # general layout only: K copies of the volume
        s = torch.cuda.Stream(device=dev)
        ren.set_stream(s.cuda_stream)
        _setup(ovr, ren, vol, n, size, 2, accumulate=True, shard=(k, K, 16, 16) if K > 1 else None)
        rens.append(ren); streams.append(s)
    for _ in range(4):
        for r in rens:
            r.render_async()
    for r in rens:
        r.sync()
    torch.cuda.synchronize()
    if offset_ms > 0.0 and K > 1:   # start the streams staggered: stream k spins k * offset first
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                torch.cuda._sleep(int(k * offset_ms * 1e-3 * 2.4e9))
    t0 = time.perf_counter()
    for _ in range(frames):
        for r in rens:
            r.render_async()
    for r in rens:
        r.sync()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames * 1e3
    ks = [r.stats().kernel_ms for r in rens]
    for r in rens:
        r.close()
    return dt, ks


for K, off in ((1, 0.0), (2, 0.0), (2, 0.75), (3, 0.5), (4, 0.0), (4, 0.4), (1, 0.0)):
    dt, ks = run(K, off)
    print(f'{K} concurrent shard renderer(s), stagger {off:.2f} ms: {dt:.3f} ms per whole frame ({1e3 / dt:.0f} fps); last frame kernel ms per renderer: ' + ' '.join(f'{k:.2f}' for k in ks), flush=True)
