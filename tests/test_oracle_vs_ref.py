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
