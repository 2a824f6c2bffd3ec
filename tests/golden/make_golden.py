"""Generates tests/golden/frames.npz: small full-frame outputs of the CPU oracle (RGBA32F, gradient layer, counters)
for a fixed set of seeded scenes.  They guard the oracle itself against regressions (the GPU parity tests compare with
the live oracle AND with these files).  The reference has no golden images, so these are oracle-generated.
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle as O  # noqa: E402
import ovr_amd as ovr  # noqa: E402
from helpers import make_case, oracle_scene  # noqa: E402

CASES = {
    "f32_full_front": dict(n=16, tf="sparse", cam="front", size=(32, 24), shading=2),
    "f32_full_oblique": dict(n=16, tf="bumps", cam="oblique", size=(32, 24), shading=2),
    "f32_grad_oblique": dict(n=16, tf="dense", cam="oblique", size=(32, 24), shading=1),
    "f32_none_front": dict(n=16, tf="sparse", cam="front", size=(32, 24), shading=0),
    "u8_full_oblique": dict(n=16, dtype=np.uint8, tf="sparse", cam="oblique", size=(32, 24), shading=2),
    "u16_full_front_rate2": dict(n=12, dtype=np.uint16, tf="bumps", cam="front", size=(24, 24), shading=2, rate=2.0),
    "f32_inside_vertex": dict(n=16, tf="sparse", cam="inside", size=(24, 24), shading=2, convention=1),
}


def generate():
    out = {}
    for name, kw in CASES.items():
        case = make_case(ovr, O, **kw)
        rgba, grad, cnt = oracle_scene(O, case).render(nthreads=1)
        out[name + "/rgba"] = rgba
        out[name + "/grad"] = grad
        out[name + "/counters"] = np.array([cnt.rays, cnt.samples, cnt.shaded_samples, cnt.shadow_samples, cnt.shadow_samples_visible], dtype=np.int64)
    # accumulation: 3 frames, spp 2 (TEA jitter)
    case = make_case(ovr, O, n=12, tf="sparse", cam="oblique", size=(24, 16), shading=2, spp=2)
    rgba, grad, cnt = oracle_scene(O, case).render(frames=3, accumulate=True, nthreads=1)
    out["accum3_spp2/rgba"] = rgba
    out["accum3_spp2/counters"] = np.array([cnt.rays, cnt.samples, cnt.shaded_samples, cnt.shadow_samples, cnt.shadow_samples_visible], dtype=np.int64)
    return out


def main():
    """frames.npz: the oracle's default restatement of __powf, exp2f(y * log2f(x)) (round 5); frames_powf_libm.npz: the same scenes with libm's powf
    (O.POWF_LIBM) - the file rounds 1-4 committed as frames.npz, kept so that the switch is known to reproduce the old oracle bit for bit"""
    for mode, name in ((O.POWF_EXP2_LOG2, "frames.npz"), (O.POWF_LIBM, "frames_powf_libm.npz")):
        old = O.set_powf_mode(mode)
        out = generate()
        O.set_powf_mode(old)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name), **out)
        print("wrote", len(out), "arrays to", name)


if __name__ == "__main__":
    main()
