"""ovr_hip_set_volume of a device array, timed call by call (the first call allocates; later ones find the freed block in the runtime's pool):
python tools/upload_time.py [edge] [dtype]"""
import sys, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R]
import torch, ovr_amd as ovr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dtype = sys.argv[2] if len(sys.argv) > 2 else "float32"
vol = ovr.synth.make_volume_torch(n, torch.device("cuda", 0), dtype)
ren = ovr.create_renderer("hip")
import numpy as np
colors, alphas, vr = ovr.synth.make_tfn("sparse", 1024, np.dtype(dtype))
ren.set_transfer_function(colors, alphas, vr)
scene = ovr.Scene(volume=vol, grid_origin=(0, 0, 0), grid_spacing=(1, 1, 1), transfer_function=None, volume_sampling_rate=1.0)
for i in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ren._upload_volume(scene)
    t1 = time.perf_counter()
    info = ren.volume_info()
    print(f"set_volume {n}^3 {dtype} call {i}: {(t1 - t0) * 1e3:.2f} ms, resident {info.resident_bytes / 1e9:.2f} GB", flush=True)
ren.close()
