"""Which combination of non-finite / huge voxels makes the HIP frame differ from the oracle's?  (diagnostic; python tests/nonfinite_diag.py)"""
import itertools
import sys
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + '/tests', _R + '/oracle']
import numpy as np
import ovr_amd as ovr
import oracle as O
from helpers import make_case, oracle_scene, hip_setup, hip_frame

SPECIAL = np.array([np.nan, np.inf, -np.inf, np.nan, 3.0e38, -3.0e38], np.float32)


def run(seed, keep, shading=1, pipeline=2):
    rng = np.random.default_rng(seed)
    dims = (14, 12, 13)
    case = make_case(ovr, O, n=14, dtype=np.float32, tf="dense", cam="oblique", size=(48, 40), shading=shading, dims=dims, tf_n=128)
    v = case["vol"]
    idx = rng.integers(0, v.size, 6)
    for k in keep:
        v.reshape(-1)[idx[k]] = SPECIAL[k]
    ref, _, cnt = oracle_scene(O, case).render()
    ren = ovr.create_renderer("hip")
    hip_setup(ovr, ren, case, pipeline=pipeline)
    ren.render()
    got = hip_frame(ovr, ren)[0]
    ren.close()
    d = np.abs(got - ref); d = np.where(np.isnan(d), np.inf, d)
    return float(d.max()), [tuple(int(c) for c in np.unravel_index(int(idx[k]), v.shape)) for k in keep], got, ref


bad = []
for seed in range(60):
    m, pos, _, _ = run(seed, range(6))
    if m > 2e-4:
        bad.append(seed)
        print(f"seed {seed}: max diff {m:.3g} voxels (z,y,x) {pos}", flush=True)
print("failing seeds:", bad)
for seed in bad[:3]:
    for r in (1, 2):
        for keep in itertools.combinations(range(6), r):
            m, pos, got, ref = run(seed, keep)
            if m > 2e-4:
                d = np.abs(got - ref); d = np.where(np.isnan(d), np.inf, d)
                y, x, ch = np.unravel_index(np.argmax(d), d.shape)
                print(f"  seed {seed} subset {[str(SPECIAL[k]) for k in keep]} at {pos}: diff {m:.3g} pixel ({x},{y}) hip {got[y, x]} oracle {ref[y, x]}", flush=True)
