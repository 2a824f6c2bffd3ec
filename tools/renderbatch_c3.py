"""The reference's OWN measurement of C3 through the drop-in boundary: the unmodified `renderbatch` (apps/main_batch.cpp, built in
place by oracle/build_ref.sh) loads plugin/libdevice_hip.so (`--device hip`), reads a VIDI3D scene whose raw volume is the bench's
synthetic 1024^3 f32 field, renders 5 + 25 blocking frames at 1920x1080 (main_batch.cpp:278-289) and prints its `fps = ...` line.
usage (on the GPU box): python tools/renderbatch_c3.py [n] [W,H] [dtype]   (writes n^3 voxels to a scratch directory under /tmp)"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ovr_amd as ovr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
fbsize = sys.argv[2] if len(sys.argv) > 2 else "1920,1080"
dtype = sys.argv[3] if len(sys.argv) > 3 else "float32"
np_dtype = {"float32": np.float32, "uint8": np.uint8, "uint16": np.uint16}[dtype]
renderbatch = os.path.join(ROOT, "oracle", "_ref", "renderbatch")
plugin = os.path.join(ROOT, "plugin", "libdevice_hip.so")
if not (os.path.exists(renderbatch) and os.path.exists(plugin)):
    raise SystemExit("oracle/_ref/renderbatch or plugin/libdevice_hip.so missing (built by __graft_entry__.build() where the reference tree is present)")

d = tempfile.mkdtemp(prefix="ovr_c3_", dir="/tmp")
try:
    t0 = time.perf_counter()
    vol = ovr.synth.make_volume_torch(n, torch.device("cuda", 0), dtype).cpu().numpy()
    if vol.dtype != np_dtype:
        vol = vol.view(np_dtype)
    torch.cuda.empty_cache()
    colors, alphas, vr = ovr.synth.make_tfn("sparse", 1024, np_dtype)
    cam = ovr.synth.make_camera("oblique", n)
    scene = ovr.vidi3d.write_scene(d, "c3", vol, ovr.synth._RAINBOW, alphas[1::2].copy(), (0.0, 1.0), cam, fovy=45.0, sample_distance=1.0)
    del vol
    print(f"[renderbatch_c3] scene written in {time.perf_counter() - t0:.1f} s: {scene}", flush=True)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.dirname(plugin), os.path.join(ROOT, "open-volume-renderer_amd"), env.get("LD_LIBRARY_PATH", "")])
    for skip in ("0", "1"):
        env["OVR_HIP_SKIP_EMPTY"] = skip   # the plugin's switch (plugin/device_hip.cpp): empty-space skipping is its default; frames are bit-identical
        t1 = time.perf_counter()
        out = subprocess.run([renderbatch, "--scene", scene, "--num-frames", "1", "--device", "hip", "--fbsize", fbsize, "--exp", os.path.join(d, "out" + skip)],
                             env=env, cwd=d, capture_output=True, text=True, timeout=900)
        if out.returncode != 0:
            print(out.stdout[-3000:], out.stderr[-3000:])
            raise SystemExit(out.returncode)
        fps = [l for l in out.stdout.splitlines() if l.startswith("fps =")]
        print(f"[renderbatch_c3] {n}^3 {dtype} at {fbsize}, empty-space skipping {'on (plugin default)' if skip == '1' else 'off'}: the reference app's own line: "
              f"{fps[-1] if fps else 'no fps line'}  (whole run incl. volume load {time.perf_counter() - t1:.1f} s)", flush=True)
    # (round 4) the same unmodified app on a device group: OVR_HIP_DEVICES makes the plugin create one (ovr_hip_create_group); argv[4] = the list
    # (default 0,0: two members on one card - the path, not the speed; on an 8-GPU node: 0,1,2,3,4,5,6,7)
    devices = sys.argv[4] if len(sys.argv) > 4 else "0,0"
    env["OVR_HIP_SKIP_EMPTY"] = "0"
    out = subprocess.run([renderbatch, "--scene", scene, "--num-frames", "1", "--device", "hip", "--fbsize", fbsize, "--exp", os.path.join(d, "outg")],
                         env=dict(env, OVR_HIP_DEVICES=devices), cwd=d, capture_output=True, text=True, timeout=900)
    fps = [l for l in out.stdout.splitlines() if l.startswith("fps =")]
    g = open(os.path.join(d, "outg000000.png"), "rb").read() if os.path.exists(os.path.join(d, "outg000000.png")) else None
    one = open(os.path.join(d, "out0000000.png"), "rb").read() if os.path.exists(os.path.join(d, "out0000000.png")) else None
    print(f"[renderbatch_c3] OVR_HIP_DEVICES={devices} (device group behind the unmodified app): {fps[-1] if fps else out.stderr[-500:]}; PNG "
          f"{'byte-identical to the one-device run' if g is not None and g == one else 'DIFFERS / missing'}", flush=True)
    # renderapp's render-thread order (commit -> mapframe -> swap -> render: every frame crosses PCIe) through the same plugin
    probe = os.path.join(ROOT, "oracle", "_ref", "plugin_probe")
    w_, h_ = fbsize.split(",")
    if os.path.exists(probe):
        env["OVR_HIP_SKIP_EMPTY"] = "0"
        for label, extra in (("RGBA + gradient layer, cropped to the box's screen rectangle (default)", {}),
                             ("RGBA only (OVR_HIP_MAP_GRAD=0), cropped", {"OVR_HIP_MAP_GRAD": "0"}),
                             ("RGBA + gradient layer, whole frame (round 2's copy)", {"OVR_HIP_MAP_WHOLE_FRAME": "1"})):
            out = subprocess.run([probe, "--loop", "100", scene, w_, h_], env=dict(env, **extra), cwd=d, capture_output=True, text=True, timeout=900)
            line = [l for l in out.stdout.splitlines() if l.startswith("loop fps")]
            print(f"[renderbatch_c3] mapped every frame, {label}: {line[-1] if line else out.stderr[-500:]}", flush=True)
    a = open(os.path.join(d, "out0000000.png"), "rb").read() if os.path.exists(os.path.join(d, "out0000000.png")) else None
    b = open(os.path.join(d, "out1000000.png"), "rb").read() if os.path.exists(os.path.join(d, "out1000000.png")) else None
    print(f"[renderbatch_c3] the two PNGs are {'byte-identical' if a is not None and a == b else 'NOT identical / missing'}")
finally:
    shutil.rmtree(d, ignore_errors=True)
