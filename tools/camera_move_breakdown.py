"""where a frame with a new camera spends its time beyond the static frame: host time of set_camera + commit and of render(), device time of the frame's
own kernels (kernel_ms), for C3's shape and a small volume; with `tf`: a transfer-function edit (same support, new values) instead of a camera move.
usage: python tools/camera_move_breakdown.py [n] [tf] [skip]"""
import sys, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + '/tests']
import numpy as np, torch, ovr_amd as ovr
from test_full_size_gpu import _setup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
edit_tf = "tf" in sys.argv[2:]
colors, alphas, vr = ovr.synth.make_tfn('sparse', 1024, np.float32)
vol = ovr.synth.make_volume_torch(n, torch.device('cuda', 0), 'float32')
ren = _setup(ovr, ovr.create_renderer('hip'), vol, n, (1920, 1080), 2, accumulate=True, skip="skip" in sys.argv[2:])
ren.set_phase_timing(False)
eye, at, up = ovr.synth.make_camera('oblique', n)
for moving in (False, True, False, True):
    for _ in range(14):
        ren.render()
    tc = tr = km = 0.0
    for i in range(50):
        t0 = time.perf_counter()
        if moving and edit_tf:
            al = alphas.copy()
            al[1::2] *= (1.0 - 0.001 * (i + 1))
            t0 = time.perf_counter()   # (the array arithmetic above is the caller's)
            ren.set_transfer_function(colors, al, vr)
            ren.commit()
        elif moving:
            a = 0.001 * (i + 1)
            ren.set_camera((eye[0] + a * n, eye[1] - a * n, eye[2]), at, up)
            ren.commit()
        t1 = time.perf_counter()
        ren.render()
        t2 = time.perf_counter()
        tc += t1 - t0; tr += t2 - t1; km += ren.stats().kernel_ms
    print(f"{n}^3 {('tf edit' if edit_tf else 'moving') if moving else 'static'}: setter + commit {tc / 50 * 1e3:.3f} ms, render() {tr / 50 * 1e3:.3f} ms (its kernels first-to-last event {km / 50:.3f} ms)")
ren.close()
