"""ms per blocking render() of C3 while the transfer function changes on every frame (an interactive TF editor): every change
uploads the tables and - with empty-space skipping, the plugin's default - rebuilds the majorant and occupancy grids.
usage: python tools/tf_edit_time.py [n]"""
import sys, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + '/tests']
import numpy as np, torch, ovr_amd as ovr
from test_full_size_gpu import _setup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
vol = ovr.synth.make_volume_torch(n, torch.device('cuda', 0), 'float32')
colors, alphas, vr = ovr.synth.make_tfn('sparse', 1024, np.float32)
for skip in (False, True):
    ren = _setup(ovr, ovr.create_renderer('hip'), vol, n, (1920, 1080), 2, accumulate=True, skip=skip)
    for editing in (False, True, False, True):
        for _ in range(5):
            ren.render()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(25):
            if editing:
                a = alphas.copy()
                a[1::2] *= (1.0 - 0.001 * (i + 1))   # same support (the same macrocells stay empty), new values
                ren.set_transfer_function(colors, a, vr)
                ren.commit()
            ren.render()
        torch.cuda.synchronize()
        print(f"skipping {'on ' if skip else 'off'} {'TF edited every frame' if editing else 'static TF           '}: {(time.perf_counter() - t0) / 25 * 1e3:.3f} ms per frame")
    ren.close()
