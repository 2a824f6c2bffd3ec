"""one-off: which layout do the frames of the front view read while / after the thin replica is built in the background?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ovr_amd as ovr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
vol = ovr.synth.make_volume_torch(n, dev, "float32")
colors, alphas, vr = ovr.synth.make_tfn("sparse", 1024, np.float32)
ren = ovr.create_renderer("hip", 0)
ren.set_fbsize((1920, 1080)); ren.set_frame_accumulation(True); ren.set_shading(2)
ren.set_transfer_function(colors, alphas, vr)
ren.init(ovr.Scene(volume=vol, transfer_function=None), ovr.Camera(*ovr.synth.make_camera("oblique", n)))
for cam in (() if "skiponly" in sys.argv else ("oblique", "front", "oblique", "front")):
    ren.set_camera(*ovr.synth.make_camera(cam, n)); ren.commit()
    seq = []
    for i in range(40):
        t0 = time.perf_counter(); ren.render(); dt = (time.perf_counter() - t0) * 1e3
        st = ren.stats()
        seq.append((st.layout, st.replicas_building, round(dt, 2), round(st.march_ms, 2), round(st.shade_ms, 2)))
    print(cam, seq[:6], "...", seq[-3:], "resident GB", ren.volume_info().resident_bytes / 1e9, flush=True)
ren.close()
# the skipping kernels: shade-heavy by the renderer's measure -> probes, the quad replica gets built
ren = ovr.create_renderer("hip", 0)
ren.set_fbsize((1920, 1080)); ren.set_frame_accumulation(True); ren.set_shading(2)
ren.set_transfer_function(colors, alphas, vr)
ren.init(ovr.Scene(volume=vol, transfer_function=None), ovr.Camera(*ovr.synth.make_camera("oblique", n)))
ren.set_empty_space_skipping(True); ren.set_camera(*ovr.synth.make_camera("oblique", n)); ren.commit()
seq = []
for i in range(70):
    t0 = time.perf_counter(); ren.render(); dt = (time.perf_counter() - t0) * 1e3
    st = ren.stats()
    seq.append((i, st.layout, st.pipeline, st.tuning, st.replicas_building, round(dt, 2), round(st.kernel_ms, 2)))
print("skip:", [s for s in seq if s[5] > 1.35 or s[3] == 1 or s[0] < 3])
print("last", seq[-3:])
