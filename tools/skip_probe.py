"""march time with empty-space skipping for: an all-transparent TF (every round skipped by the ray interval), the bench TF"""
import sys
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + '/tests']
import numpy as np, torch, ovr_amd as ovr
n, size = 1024, (1920, 1080)
vol = ovr.synth.make_volume_torch(n, torch.device('cuda', 0), 'float32')
for name in ('empty', 'sparse'):
    colors, alphas, vr = ovr.synth.make_tfn('sparse', 1024)
    if name == 'empty':
        alphas = np.array(alphas, dtype=np.float32).copy(); alphas[1::2] = 0.0
    for skip in (False, True):
        ren = ovr.create_renderer('hip')
        ren.set_fbsize(size); ren.set_shading(2); ren.set_frame_accumulation(True); ren.set_empty_space_skipping(skip)
        ren.set_transfer_function(colors, alphas, vr)
        ren.init(ovr.Scene(volume=vol, transfer_function=None), ovr.Camera(*ovr.synth.make_camera('oblique', n)))
        ren.commit()
        best = None
        for _ in range(12):
            ren.render(); st = ren.stats()
            if best is None or st.kernel_ms < best[0]: best = (st.kernel_ms, st.march_ms, st.shade_ms, st.composite_ms)
        print(name, 'skip' if skip else 'plain', 'kernel %.3f march %.3f shade %.3f comp %.3f' % best, 'samples', st.samples, 'skipped', st.skipped_samples)
        ren.close()
