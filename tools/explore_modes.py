"""Frame times of C3 (1024^3 f32, 1920x1080, oblique camera, sparse TF) over the renderer's other switches - samples per pixel,
sampling rate, shading mode, sparse sampling, empty-space skipping - to spot modes that cost more than their work explains.
usage: python tools/explore_modes.py [n]"""
import itertools, sys, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + '/tests']
import numpy as np, torch, ovr_amd as ovr
from test_full_size_gpu import _setup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
vol = ovr.synth.make_volume_torch(n, torch.device('cuda', 0), 'float32')
ren = _setup(ovr, ovr.create_renderer('hip'), vol, n, (1920, 1080), 2, accumulate=True)
ren.set_noise_tile(ovr.synth.make_noise_tile(64))
ren.set_focus((0.5, 0.5), 0.06, 0.07)


def measure(label, frames=8):
    ren.commit()
    for _ in range(3):
        ren.render()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        ren.render()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames * 1e3
    st = ren.stats()
    print(f"{label:58s} {dt:8.3f} ms  march {st.march_ms:7.3f} shade {st.shade_ms:7.3f} comp {st.composite_ms:6.3f}  samples {st.samples / 1e6:7.1f}M shaded {st.shaded_samples / 1e6:6.1f}M "
          f"shadow {st.shadow_samples / 1e6:7.1f}M skipped {st.skipped_samples / 1e6:6.1f}M pipeline {st.pipeline} skipping_kernels {st.skipping_kernels}", flush=True)


for shading, spp, rate in itertools.product((2, 1, 0), (1, 2, 4), (1.0, 0.5, 2.0)):
    if (spp > 1 and rate != 1.0):
        continue
    ren.set_shading(shading); ren.set_sample_per_pixel(spp); ren.set_volume_sampling_rate(rate)
    measure(f"shading {shading} spp {spp} rate {rate}")
ren.set_shading(2); ren.set_sample_per_pixel(1); ren.set_volume_sampling_rate(1.0)
for sparse, skip in itertools.product((False, True), (False, True)):
    ren.set_sparse_sampling(sparse); ren.set_empty_space_skipping(skip)
    measure(f"sparse sampling {sparse} skipping {skip}")
ren.set_sparse_sampling(False); ren.set_empty_space_skipping(False)
for pipeline in (1, 2):
    ren.set_shading_pipeline(pipeline)
    measure(f"shading pipeline {pipeline} (1 = in place, 2 = pooled)", frames=3)
ren.close()
