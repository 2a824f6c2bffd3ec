"""shared helpers for the parity tests: build the same scene for the HIP device and for the CPU oracle"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(argv, env=None, timeout=900):
    """`python bench.py <argv>` as the driver runs it: stdout must carry exactly ONE line starting with `{` - the compact record, below 4 KB
    (a 20 KB line left the driver's record unparsed) - and the full record goes to --detail-file.  Returns (compact, detail, process)."""
    fd, detail = tempfile.mkstemp(prefix="ovr_bench_test_", suffix=".json")
    os.close(fd)
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv) + ["--detail-file", detail], env=env, capture_output=True, text=True, timeout=timeout)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, f"{len(lines)} JSON lines on stdout"
        assert len(lines[0].encode()) < 4096, len(lines[0])
        with open(detail) as f:
            return json.loads(lines[0]), json.load(f), out
    finally:
        try:
            os.unlink(detail)
        except OSError:
            pass


def make_case(ovr, O, n=32, dtype=np.float32, tf="sparse", cam="front", size=(64, 48), shading=2, rate=1.0, spp=1,
              convention=0, dims=None, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), fovy=60.0, tf_n=1024):
    vol = ovr.synth.make_volume(n, dtype, dims=dims)
    colors, alphas, vr = ovr.synth.make_tfn(tf, tf_n, dtype)
    nmax = max(vol.shape)
    eye, at, up = ovr.synth.make_camera(cam, nmax, "vertex" if convention else "cell")
    # scale the camera with the world-space extent of the grid
    ext = np.array(spacing, dtype=np.float64)
    eye = tuple(np.array(origin) + np.array(eye) * ext)
    at = tuple(np.array(origin) + np.array(at) * ext)
    return dict(vol=vol, colors=colors, alphas=alphas, vr=vr, cam=(eye, at, up), size=size, shading=shading, rate=rate,
                spp=spp, convention=convention, spacing=spacing, origin=origin, fovy=fovy)


def oracle_scene(O, case, **kw):
    w, h = case["size"]
    return O.OracleScene(case["vol"], case["colors"], case["alphas"], case["vr"], case["cam"], w, h, fovy=case["fovy"],
                         spp=case["spp"], rate=case["rate"], shading=case["shading"], grid_origin=case["origin"],
                         grid_spacing=case["spacing"], convention=case["convention"], **kw)


def hip_setup(ovr, ren, case, accumulate=False, pipeline=0):
    """the call sequence of the reference's renderbatch (apps/main_batch.cpp:254-276)"""
    scene = ovr.Scene(volume=case["vol"], grid_origin=case["origin"], grid_spacing=case["spacing"],
                      transfer_function=None, volume_sampling_rate=case["rate"])
    eye, at, up = case["cam"]
    ren.set_fbsize(case["size"])
    ren.set_frame_accumulation(accumulate)
    ren.set_sample_per_pixel(case["spp"])
    ren.set_volume_sampling_rate(case["rate"])
    ren.set_shading(case["shading"])
    ren.set_shading_pipeline(pipeline)
    ren.set_grid_convention(case["convention"])
    ren.set_transfer_function(case["colors"], case["alphas"], case["vr"])
    ren.init(scene, ovr.Camera(eye, at, up, case["fovy"]))
    ren.set_sparse_sampling(False)
    ren.commit()
    return ren


def hip_frame(ovr, ren):
    fb = ovr.FrameBufferData()
    ren.mapframe(fb)
    return np.array(fb.rgba.data(), copy=True), np.array(fb.grad.data(), copy=True)


# OVR_PARITY_EXACT_RUN=1 (tests/test_parity_exact_gpu.py starts the parity suites that way, with OVR_HIP_LIBRARY = the exact-parity build of the kernels and the
# oracle in its "det" mode): the parity bar below becomes EQUALITY - every float of the frame, bit for bit
EXACT_RUN = os.environ.get("OVR_PARITY_EXACT_RUN") == "1"


def compare(O, rgba_hip, rgba_ref, tol_float=2e-4, name=""):
    """the parity bar: <= 1 on every 8-bit channel after the reference's only "tonemap" (imageio.cpp:146-181),
    plus a float tolerance that is far tighter than that"""
    if EXACT_RUN:
        same = np.asarray(rgba_hip, dtype=np.float32).view(np.uint32) == np.asarray(rgba_ref, dtype=np.float32).view(np.uint32)
        assert same.all(), f"{name}: exact-parity run: {int((~same).sum())} of {same.size} floats differ from the oracle's (max {np.abs(rgba_hip - rgba_ref).max()})"
        return 0, 0.0
    a8, b8 = O.rgba8(rgba_hip), O.rgba8(rgba_ref)
    d8 = np.abs(a8.astype(np.int32) - b8.astype(np.int32)).max()
    df = np.abs(rgba_hip - rgba_ref).max()
    assert not np.isnan(rgba_hip).any(), f"{name}: NaN in HIP frame"
    assert d8 <= 1, f"{name}: 8-bit channel difference {d8} > 1 (float diff {df})"
    assert df <= tol_float, f"{name}: float difference {df} > {tol_float}"
    return d8, df
