"""where a frame with a new camera spends its time beyond the static frame: host time of set_camera + commit and of render(), device time of the frame's
own kernels (kernel_ms), for C3's shape and a small volume.   usage: python tools/camera_move_breakdown.py [n]"""
import sys, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + '/tests']
import numpy as np, torch, ovr_amd as ovr
from test_full_size_gpu import _setup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
vol = ovr.synth.make_volume_torch(n, torch.device('cuda', 0), 'float32')
ren = _setup(ovr, ovr.create_renderer('hip'), vol, n, (1920, 1080), 2, accumulate=True)
ren.set_phase_timing(False)
eye, at, up = ovr.synth.make_camera('oblique', n)
for moving in (False, True, False, True):
    for _ in range(14):
        ren.render()
    tc = tr = km = 0.0
    for i in range(50):
        t0 = time.perf_counter()
        if moving:
            a = 0.001 * (i + 1)
            ren.set_camera((eye[0] + a * n, eye[1] - a * n, eye[2]), at, up)
            ren.commit()
        t1 = time.perf_counter()
        ren.render()
        t2 = time.perf_counter()
        tc += t1 - t0; tr += t2 - t1; km += ren.stats().kernel_ms
    print(f"{n}^3 {'moving' if moving else 'static'}: set_camera + commit {tc / 50 * 1e3:.3f} ms, render() {tr / 50 * 1e3:.3f} ms (its kernels first-to-last event {km / 50:.3f} ms)")
ren.close()
