"""Pins the oracle's host-side restatements against REAL reference code.

tests/golden/ref_probe.json was produced by oracle/_ref/ref_probe (oracle/ref_probe.cpp compiled against the reference's own
headers and objects by oracle/build_ref.sh; `make -C oracle golden`).  It holds outputs of the reference's image_to_rgba8
(the only "tonemap"), of the camera-basis formulas evaluated with the reference's gdt math, and of gdt's affine transforms."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "ref_probe.json")) as f:
    REF = json.load(f)


def test_rgba8_matches_reference_bit_exact(oracle):
    w, h = map(int, REF["rgba8_dims"])
    img = np.array(REF["rgba8_input"], dtype=np.float32).reshape(h, w, 4)
    for flip, key in ((False, "rgba8_plain"), (True, "rgba8_flipped")):
        ref = np.array(REF[key], dtype=np.uint8).reshape(h, w, 4)
        got = oracle.rgba8(img, flip=flip)
        assert np.array_equal(got, ref), key


def test_camera_basis_matches_gdt(oracle):
    cin = np.array(REF["camera_input"], dtype=np.float64).reshape(-1, 12)
    cout = np.array(REF["camera_basis"], dtype=np.float32).reshape(-1, 12)
    for a, ref in zip(cin, cout):
        got = oracle.camera_basis(a[0:3], a[3:6], a[6:9], a[9], int(a[10]), int(a[11]))
        # the oracle writes dot() as an fma chain (what nvcc contracts to), host gdt does not: allow 2 ulp
        assert np.allclose(got, ref, rtol=3e-7, atol=1e-7), (got, ref)


def test_affine_transforms_match_gdt(oracle):
    o_s = REF["xfm_origin_scale"]
    pts = np.array(REF["xfm_points"], dtype=np.float64).reshape(-1, 3)
    ref = np.array(REF["xfm_results"], dtype=np.float32).reshape(-1, 12)
    for p, r in zip(pts, ref):
        got = oracle.xfm_probe(o_s[0:3], o_s[3:6], p)
        # gdt inverts through adjoint/det; the oracle restates the diagonal inverse as 1/s: 2 ulp
        assert np.allclose(got, r, rtol=4e-7, atol=1e-7), (got, r)


def test_value_type_numbering_is_the_abi(ovr):
    vt = np.array(REF["value_types"]).reshape(-1, 2)
    L = ovr._lib
    mine = {L.TYPE_UINT8: 1, L.TYPE_INT8: 1, L.TYPE_UINT16: 2, L.TYPE_INT16: 2, L.TYPE_UINT32: 4, L.TYPE_INT32: 4, L.TYPE_FLOAT: 4, L.TYPE_DOUBLE: 8}
    assert {int(a): int(b) for a, b in vt} == mine


def _exr_golden():
    w, h = int(REF["exr_dims"][0]), int(REF["exr_dims"][1])
    fin = np.array(REF["exr_input_bits"], dtype=np.uint32).view(np.float32).reshape(h, w, 4)
    out = np.array(REF["exr_roundtrip_bits"], dtype=np.uint32).reshape(h, w, 4)
    return fin, out


def test_exr_half_conversion_matches_reference_bit_exact(oracle):
    """ovr::save_image("*.exr") + load_exr of the REAL reference on 320 values (ties between halves, subnormal halves, float
    denormals, the overflow tie 65520, carries into the exponent): the oracle's float -> half restatement gives the same bits.
    The reference's writer also rotates the channels (R <- G, G <- B, B <- A, A <- R; imageio.cpp:27-62) - restated, not fixed."""
    fin, out = _exr_golden()
    lib = oracle.load()
    half = oracle.float_to_half(fin)
    back = np.array([lib.ovr_oracle_half_to_float(int(v)) for v in half.ravel()], dtype=np.float32).view(np.uint32).reshape(half.shape)
    assert np.array_equal(back[..., [1, 2, 3, 0]], out)
    assert np.isinf(out.view(np.float32)).any() and (out.view(np.float32) == 0).any()


def test_own_exr_writer_is_read_by_the_reference_loader(ovr, oracle, tmp_path):
    """a file written by open-volume-renderer_amd/imageio.py (ZIP and uncompressed) is loaded by the reference's tinyexr (LoadEXR in
    oracle/_ref/libovr_refhost.so) to exactly the floats the reference's own EXR of the same image holds"""
    import ctypes as C
    import pytest
    lib_path = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libovr_refhost.so")
    if not os.path.exists(lib_path):
        pytest.skip("reference host library not built (needs the reference tree at build time)")
    ref = C.CDLL(lib_path)
    ref.LoadEXR.argtypes = [C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.POINTER(C.c_char_p)]
    fin, out = _exr_golden()
    h, w = fin.shape[:2]
    half = oracle.float_to_half(fin)[::-1]   # save_image flips the rows before the write (imageio.cpp:271)
    for comp in ("zip", "none"):
        path = str(tmp_path / f"own_{comp}.exr")
        ovr.imageio.save_exr(path, half, compression=comp)
        data, ww, hh, err = C.POINTER(C.c_float)(), C.c_int(), C.c_int(), C.c_char_p()
        rc = ref.LoadEXR(C.byref(data), C.byref(ww), C.byref(hh), path.encode(), C.byref(err))
        assert rc == 0, err.value
        assert (ww.value, hh.value) == (w, h)
        got = np.ctypeslib.as_array(data, shape=(h, w, 4)).view(np.uint32)[::-1]
        assert np.array_equal(got, out), comp
    # named for what they hold, the channels come back unrotated
    path = str(tmp_path / "plain.exr")
    ovr.imageio.save_exr(path, half, reference_channel_naming=False)
    data = C.POINTER(C.c_float)()
    assert ref.LoadEXR(C.byref(data), C.byref(ww), C.byref(hh), path.encode(), C.byref(err)) == 0
    got = np.ctypeslib.as_array(data, shape=(h, w, 4)).view(np.uint32)[::-1]
    assert np.array_equal(got[..., [1, 2, 3, 0]], out)


# ---- the same pins on WIDE seeded inputs (tests/golden/ref_probe_wide.npz, generated by tests/golden/make_ref_probe_wide.py through
# ---- oracle/_ref/ref_probe_wide: the reference's own compiled functions) ------------------------------------------------------------
WIDE = np.load(os.path.join(HERE, "golden", "ref_probe_wide.npz"))
_f = lambda k: WIDE[k].view(np.float32)


def test_rgba8_matches_reference_bit_exact_wide(oracle):
    """every float within 6 ulp of k / 255 for all 256 k (where the truncating quantiser changes its answer), zeros, denormals, huge values,
    infinities and 4096 random floats in [-0.5, 1.5]"""
    img = _f("rgba8_in").reshape(1, -1, 4)
    ref = WIDE["rgba8_out"].reshape(1, -1, 4)
    assert np.array_equal(oracle.rgba8(img, flip=False), ref)
    assert len(np.unique(ref)) == 256


def test_camera_basis_matches_gdt_wide(oracle):
    """400 random cameras at four scales (positions of order 0.01 ... 1000), random and axis-aligned up vectors, fovy 5 ... 120, any frame size"""
    cin, cout = _f("camera_in").reshape(-1, 12).astype(np.float64), _f("camera_out").reshape(-1, 12)
    worst = 0.0
    for a, ref in zip(cin, cout):
        got = np.asarray(oracle.camera_basis(a[0:3], a[3:6], a[6:9], a[9], int(a[10]), int(a[11])), dtype=np.float32)
        # relative to the length of each basis vector (components near 0 carry the vector's absolute rounding error)
        for k in range(0, 12, 3):
            scale = max(float(np.linalg.norm(ref[k:k + 3])), 1e-30)
            worst = max(worst, float(np.abs(got[k:k + 3] - ref[k:k + 3]).max()) / scale)
    assert worst <= 6e-7, worst   # the oracle writes dot() as an fma chain (what nvcc contracts to), host gdt does not


def test_affine_transforms_match_gdt_wide(oracle):
    """200 random instance transforms translate(origin) * scale(s), s from 0.05 to 3000 per axis, and points"""
    ain, aout = _f("affine_in").reshape(-1, 9).astype(np.float64), _f("affine_out").reshape(-1, 12)
    worst = 0.0
    for a, ref in zip(ain, aout):
        got = np.asarray(oracle.xfm_probe(a[0:3], a[3:6], a[6:9]), dtype=np.float32)
        for k in range(0, 12, 3):
            scale = max(float(np.abs(ref[k:k + 3]).max()), 1e-30)
            worst = max(worst, float(np.abs(got[k:k + 3] - ref[k:k + 3]).max()) / scale)
    assert worst <= 1e-6, worst   # gdt inverts through adjoint / det; the oracle restates the diagonal inverse as 1 / s


def test_exr_half_conversion_matches_reference_bit_exact_wide(oracle):
    """16 384 floats over the whole half range and beyond - exponents 2^-27 ... 2^17, a quarter of them exact ties between two halves,
    others one bit to either side of a tie - through the reference's save_image("*.exr") + load_exr: same bits from the oracle's rule"""
    fin = _f("exr_in").reshape(4, 1024, 4)
    out = WIDE["exr_out"].reshape(4, 1024, 4)
    lib = oracle.load()
    half = oracle.float_to_half(fin)
    back = np.array([lib.ovr_oracle_half_to_float(int(v)) for v in half.ravel()], dtype=np.float32).view(np.uint32).reshape(half.shape)
    assert np.array_equal(back[..., [1, 2, 3, 0]], out)
    o = out.view(np.float32)
    assert np.isinf(o).any() and (o == 0).any() and ((np.abs(o) > 0) & (np.abs(o) < 6.2e-5)).any()   # overflow, underflow and subnormal halves all occur


def test_every_literal_of_the_integration_loop_is_the_references(oracle):
    """tests/golden/ref_literals.json holds the numeric literals of the reference's integration loop as its source text spells them (ERT thresholds,
    shadow step scale, midpoint factor, nearly_equal's epsilon, the light, the shading terms, TEA's rounds / constants / scale, float_small / float_large),
    extracted from the cited lines by tests/golden/make_ref_literals.py.  The restatement's constants (ovr_oracle_literals) must be those values as float32 /
    uint32.  A pin of the CONSTANTS, not of the arithmetic around them: the oracle stays "parity unpinned" for that."""
    with open(os.path.join(HERE, "golden", "ref_literals.json")) as f:
        ref = json.load(f)
    mine = oracle.literals()
    alias = {"midpoint_shadow": "midpoint"}  # the shadow march (:67) and the primary march (:112) spell the same factor; the restatement has one
    unused = {"pixel_jitter_centre", "screen_centre_x", "screen_centre_y"}  # checked below against the numbers the raygen restatement uses
    for name, rec in ref.items():
        if name in unused:
            assert rec["value"] == 0.5, (name, rec)
            continue
        got = mine[alias.get(name, name)]
        if rec["kind"] == "int":
            assert int(got) == int(rec["value"]), (name, got, rec)
        else:
            assert np.float32(got) == np.float32(rec["value"]), (name, got, rec)
    assert set(mine) <= set(ref) | set(alias.values()), set(mine) - set(ref)
