"""End-to-end drop-in check: the UNMODIFIED reference app (oracle/_ref/renderbatch = apps/main_batch.cpp built in place by
oracle/build_ref.sh) loads plugin/libdevice_hip.so through the reference's own factory (`--device hip`), renders a scene
JSON on the MI355X and writes its PNG; the PNG must match the CPU oracle's frame after the reference's 8-bit quantisation."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RENDERBATCH = os.path.join(ROOT, "oracle", "_ref", "renderbatch")
PLUGIN = os.path.join(ROOT, "plugin", "libdevice_hip.so")


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
def test_renderbatch_device_hip(tmp_path, ovr, oracle, dtype):
    if not (os.path.exists(RENDERBATCH) and os.path.exists(PLUGIN)):
        pytest.skip("oracle/_ref/renderbatch or plugin/libdevice_hip.so missing: they are built by __graft_entry__.build() where the reference tree is present and travel with the snapshot")
    from PIL import Image
    n, W, H = 40, 96, 64
    vol = ovr.synth.make_volume(n, dtype)
    colors, alphas, vr = ovr.synth.make_tfn("bumps", 256, dtype)
    alpha_table = alphas[1::2].copy()
    cam = ovr.synth.make_camera("oblique", n)
    scene = ovr.vidi3d.write_scene(str(tmp_path), "synthetic", vol, ovr.synth._RAINBOW, alpha_table, (0.0, 1.0), cam, fovy=45.0,
                                   sample_distance=0.25)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.dirname(PLUGIN), os.path.join(ROOT, "open-volume-renderer_amd"), env.get("LD_LIBRARY_PATH", "")])
    out = subprocess.run([RENDERBATCH, "--scene", scene, "--num-frames", "1", "--device", "hip", "--fbsize", f"{W},{H}",
                          "--exp", str(tmp_path / "out")], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "fps =" in out.stdout
    png = np.asarray(Image.open(str(tmp_path / "out000000.png")).convert("RGBA"))
    assert png.shape == (H, W, 4)
    # what renderbatch actually asks for: JSON camera position but fovy 60 (renderer.h:149-152), rate 1 (main_batch.cpp:69),
    # spp 1, accumulation on over 5 + 25 identical frames, reference shading; the serializer zeroes end alphas < 0.01
    alpha_ref = alpha_table.copy()
    if alpha_ref[0] < 0.01:
        alpha_ref[0] = 0.0
    if alpha_ref[-1] < 0.01:
        alpha_ref[-1] = 0.0
    alphas_ref = alphas.copy()
    alphas_ref[1::2] = alpha_ref
    sc = oracle.OracleScene(vol, ovr.synth.rainbow_colors(256), alphas_ref, vr, cam, W, H, fovy=60.0, rate=1.0, shading=oracle.SHADE_FULL)
    ref, _, _ = sc.render()
    ref8 = oracle.rgba8(ref, flip=True)
    d = np.abs(png.astype(int) - ref8.astype(int))
    # the reference rasterises the colour control points itself (tfn::updateColorMap); np.interp differs from it in the last
    # ulp, hence <= 1 on the 8-bit channels
    assert d.max() <= 1, f"max 8-bit difference {d.max()}"
    assert (ref8[..., 3] > 0).mean() > 0.02
