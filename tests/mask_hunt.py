"""One-off hunt: the sparse-sampling mask (generate_mask.cu:55-120 restated) for random frame sizes up to 4K, noise tiles, focus windows, base noise
and frame indices - the device's compacted pixel list against the oracle's, bit for bit.   usage: python tests/mask_hunt.py [cases] [seed]"""
import sys
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + '/tests', _R + '/oracle']
import numpy as np
import ovr_amd as ovr
import oracle as O
from helpers import make_case, hip_setup

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
case = make_case(ovr, O, n=8, tf="dense", cam="front", size=(64, 64), shading=0)
ren = hip_setup(ovr, ovr.create_renderer("hip"), case)
bad = 0
for i in range(cases):
    xy = int(rng.choice([16, 32, 64, 128]))
    noise = rng.random((xy, xy, 64), dtype=np.float32) if rng.integers(2) else (rng.integers(0, 256, size=(xy, xy, 64)) / 255.0).astype(np.float32)
    big = rng.integers(6) == 0
    size = (int(rng.integers(1, 3841 if big else 700)), int(rng.integers(1, 2161 if big else 500)))
    focus = ((float(rng.uniform(-0.2, 1.2)), float(rng.uniform(-0.2, 1.2))), float(10 ** rng.uniform(-2, 0.5)), float(rng.choice([0.0, 0.01, 0.1, 0.5, 1.0, rng.uniform(0, 1)])))
    ren.set_fbsize(size); ren.set_noise_tile(noise); ren.set_focus(*focus); ren.commit()
    for frame in (int(rng.integers(1, 400)), 64, 1):
        got, exp = ren.sparse_mask(frame), O.sparse_mask(frame, size[0], size[1], focus[0], focus[1], focus[2], noise)
        if not np.array_equal(got, exp):
            bad += 1
            print(f"case {i}: size {size} tile {xy} focus {focus} frame {frame}: device {len(got) // 2} pixels, oracle {len(exp) // 2}", flush=True)
            break
ren.close()
print(f"{cases} mask configurations, {bad} differ")
