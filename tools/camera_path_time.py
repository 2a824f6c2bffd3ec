"""ms per blocking render() of C3 with a static camera and with a camera that moves on every frame (every commit of a new camera rebuilds the
launch order of the march: schedule kernels; accumulation restarts like in the reference).  Round 3: also the regime the interactive app runs
the shipped scenes in - the scene files' sampling rate 4 with a dense transfer function - where a moving camera never sits still long enough
for the renderer to measure layout and pipeline: the decision measured while the camera rested is kept while it moves.
usage: python tools/camera_path_time.py [n] [rate] [tf]"""
import sys, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + '/tests']
import numpy as np, torch, ovr_amd as ovr
from test_full_size_gpu import _setup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tf = sys.argv[3] if len(sys.argv) > 3 else 'sparse'
vol = ovr.synth.make_volume_torch(n, torch.device('cuda', 0), 'float32')
ren = _setup(ovr, ovr.create_renderer('hip'), vol, n, (1920, 1080), 2, accumulate=True)
ren.set_volume_sampling_rate(rate)
ren.set_transfer_function(*ovr.synth.make_tfn(tf, 1024, np.float32))
ren.commit()
eye, at, up = ovr.synth.make_camera('oblique', n)
for moving in (False, True, False, True):
    for _ in range(14):
        ren.render()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    km, seen = 0.0, set()
    for i in range(25):
        if moving:
            a = 0.002 * (i + 1)
            e = (eye[0] + a * n, eye[1] - a * n, eye[2])
            ren.set_camera(e, at, up)
            ren.commit()
        ren.render()
        st = ren.stats()
        km += st.kernel_ms
        seen.add((st.layout, st.pipeline, st.tuning))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 25 * 1e3
    print(f"rate {rate} {tf}: {'moving' if moving else 'static'} camera: {dt:.3f} ms per frame (kernels {km / 25:.3f} ms), (layout, pipeline, tuning) seen: {sorted(seen)}")
ren.close()
