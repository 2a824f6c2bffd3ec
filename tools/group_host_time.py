"""Host time per frame of the in-process device group (ovr_hip_create_group) with n members - a rehearsal on ONE card (device 0 listed n times: the
members share the GPU, so device times mean nothing; what is measured is the leader thread's time to get every member's frame launched and
shipped).  Round 4 (one thread drives all members): 23 / 90 / 225 us at 2 / 4 / 8 members.   python tools/group_host_time.py [edge] [members ...]"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R]
import numpy as np, torch
import ovr_amd as ovr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
members = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
vol = ovr.synth.make_volume_torch(n, torch.device("cuda", 0), "float32")
colors, alphas, vr = ovr.synth.make_tfn("sparse", 1024, np.float32)
cam = ovr.synth.make_camera("oblique", n)
for k in members:
    ren = ovr.create_renderer("hip", 0, devices=[0] * k if k > 1 else None)
    ren.set_fbsize((1920, 1080)); ren.set_frame_accumulation(True); ren.set_shading(2); ren.set_transfer_function(colors, alphas, vr)
    ren.set_phase_timing(False)
    ren.init(ovr.Scene(volume=vol, transfer_function=None), ovr.Camera(*cam)); ren.set_camera(*cam); ren.commit()
    for _ in range(8):
        ren.render()
    acc = np.zeros(4); wall = 0.0; frames = 40
    for _ in range(frames):
        t0 = time.perf_counter(); ren.render(); wall += time.perf_counter() - t0
        acc += np.array(ren.group_host_times())
    st = ren.stats()
    # a moving camera: commit + render per frame (every member re-classifies its blocks and reads two words back - on its own thread)
    mv = 0.0
    for i in range(frames):
        e = np.array(cam[0]) + np.array([0.5 * i, 0.0, 0.0])
        t0 = time.perf_counter(); ren.set_camera(tuple(e), cam[1], cam[2]); ren.commit(); ren.render(); mv += time.perf_counter() - t0
    acc /= frames
    print(f"members {k}: render() {wall / frames * 1e3:.3f} ms (slowest member's kernels {st.kernel_ms:.3f} ms); host us per frame: enqueue {acc[0]:.1f} ship {acc[1]:.1f} "
          f"finish(wait) {acc[2]:.1f} scatter {acc[3]:.1f}; moving camera: commit + render {mv / frames * 1e3:.3f} ms", flush=True)
    ren.close()
