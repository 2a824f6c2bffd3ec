"""ms per blocking render() of C3 with a static camera and with a camera that moves on every frame (every commit of a new camera
rebuilds the launch order of the march: schedule kernels; accumulation restarts like in the reference).  usage: python tools/camera_path_time.py [n]"""
import sys, time
sys.path[:0] = ['/root/repo', '/root/repo/tests']
import numpy as np, torch, ovr_amd as ovr
from test_full_size_gpu import _setup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
vol = ovr.synth.make_volume_torch(n, torch.device('cuda', 0), 'float32')
ren = _setup(ovr, ovr.create_renderer('hip'), vol, n, (1920, 1080), 2, accumulate=True)
eye, at, up = ovr.synth.make_camera('oblique', n)
for moving in (False, True, False, True):
    for _ in range(5):
        ren.render()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    km = 0.0
    for i in range(25):
        if moving:
            a = 0.002 * (i + 1)
            e = (eye[0] + a * n, eye[1] - a * n, eye[2])
            ren.set_camera(e, at, up)
            ren.commit()
        ren.render()
        km += ren.stats().kernel_ms
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 25 * 1e3
    print(f"{'moving' if moving else 'static'} camera: {dt:.3f} ms per frame (march + shade + composite kernels {km / 25:.3f} ms)")
ren.close()
