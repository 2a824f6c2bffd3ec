"""One-off hunt (CPU, where oracle/_ref is built): EXR files written by open-volume-renderer_amd/imageio.py - random sizes from 1 x 1 to 300 x 200, ZIP and
uncompressed, uniform values, arbitrary bit patterns and 15 decades of magnitudes - loaded by the REFERENCE's tinyexr (LoadEXR in oracle/_ref/libovr_refhost.so):
the floats it returns must be the halves that were written, bit for bit.   usage: python tests/exr_hunt.py"""
import sys, os, ctypes as C, tempfile
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0]=[_R,_R + '/oracle']
import numpy as np
import ovr_amd as ovr
import oracle as O
ref = C.CDLL(_R + '/oracle/_ref/libovr_refhost.so')
ref.LoadEXR.argtypes = [C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.POINTER(C.c_char_p)]
lib = O.load()
rng = np.random.default_rng(3)
bad = 0
with tempfile.TemporaryDirectory() as d:
    for i in range(120):
        w, h = int(rng.integers(1, 300)), int(rng.integers(1, 200))
        kind = int(rng.integers(3))
        if kind == 0:
            img = rng.uniform(-2, 2, (h, w, 4)).astype(np.float32)
        elif kind == 1:
            bits = rng.integers(0, 2**32, (h, w, 4), dtype=np.uint64).astype(np.uint32)
            img = bits.view(np.float32)
            img = np.where(np.isnan(img), np.float32(1.5), img)     # NaN payloads are not comparable bit for bit
        else:
            img = (10.0 ** rng.uniform(-9, 6, (h, w, 4))).astype(np.float32) * rng.choice([-1, 1], (h, w, 4))
        half = O.float_to_half(img)
        for comp in ("zip", "none"):
            path = os.path.join(d, f"t_{i}_{comp}.exr")
            ovr.imageio.save_exr(path, half, compression=comp, reference_channel_naming=False)
            data, ww, hh, err = C.POINTER(C.c_float)(), C.c_int(), C.c_int(), C.c_char_p()
            rc = ref.LoadEXR(C.byref(data), C.byref(ww), C.byref(hh), path.encode(), C.byref(err))
            if rc != 0 or (ww.value, hh.value) != (w, h):
                bad += 1; print("load failed", i, comp, w, h, rc, err.value); continue
            got = np.ctypeslib.as_array(data, shape=(h, w, 4)).view(np.uint32).copy()
            exp = np.array([lib.ovr_oracle_half_to_float(int(v)) for v in half.ravel()], dtype=np.float32).view(np.uint32).reshape(h, w, 4)
            if not np.array_equal(got, exp):
                bad += 1; print("mismatch", i, comp, w, h, int((got != exp).sum()))
            os.remove(path)
print("120 images x 2 compressions,", bad, "bad")
