"""Known-answer tests of the CPU oracle (oracle/ovr_oracle.c) - the vectors SURVEY.md 8c lists, since the reference itself
holds none: TEA, box intersection, trilinear at voxel centres/corners, TF nodal lerp, opacity correction, integer
normalisation, the sparse-sampling mask, and the committed golden frames."""
import os

import numpy as np
import pytest

from helpers import make_case, oracle_scene
from make_golden import CASES

HERE = os.path.dirname(os.path.abspath(__file__))


def tea_py(v0, v1):
    """independent restatement of RandomTEA::get_floats (reference ovr/common/random/random.h:146-188)"""
    M = 0xFFFFFFFF
    s = 0
    for _ in range(16):
        s = (s + 0x9E3779B9) & M
        v0 = (v0 + ((((v1 << 4) & M) + 0xA341316C) & M ^ ((v1 + s) & M) ^ (((v1 >> 5) + 0xC8013EA4) & M))) & M
        v1 = (v1 + ((((v0 << 4) & M) + 0xAD90777D) & M ^ ((v0 + s) & M) ^ (((v0 >> 5) + 0x7E95761E) & M))) & M
    f = np.float32(2.3283064365386962890625e-10)
    return (np.float32(v0) * f, np.float32(v1) * f), (v0, v1)


@pytest.mark.parametrize("seed", [(1, 0), (1, 12345), (7, 2073599), (0xFFFFFFFF, 0xFFFFFFFF), (0, 0)])
def test_tea_known_answers(oracle, seed):
    (f0, f1), st = oracle.tea_floats(*seed)
    (g0, g1), st2 = tea_py(*seed)
    assert st == st2
    assert f0 == g0 and f1 == g1
    assert 0.0 <= f0 <= 1.0 and 0.0 <= f1 <= 1.0


def test_tea_first_values_are_stable(oracle):
    # pinned once from the two independent implementations above
    assert oracle.tea_floats(1, 0)[1] == (2376512273, 770940544)
    assert oracle.tea_floats(1, 12345)[1] == (3471014342, 405085722)


def test_box_intersection(oracle):
    FMAX = 3.4028234663852886e38
    # axis parallel through the centre
    hit, t0, t1 = oracle.intersect_box((0.5, 0.5, -1.0), (0.0, 0.0, 1.0))
    assert hit and t0 == 1.0 and t1 == 2.0
    # origin inside: t0 stays at tmin = 0
    hit, t0, t1 = oracle.intersect_box((0.25, 0.5, 0.5), (1.0, 0.0, 0.0))
    assert hit and t0 == 0.0 and t1 == 0.75
    # miss
    hit, _, _ = oracle.intersect_box((2.0, 2.0, -1.0), (0.01, 0.01, 1.0))
    assert not hit
    # reference quirk (shaders_common.h:162-172): an axis whose direction component is below FLT_MIN is ignored altogether,
    # so an exactly axis-parallel ray OUTSIDE the box in x and y still reports a hit - restated, not "fixed"
    hit, t0, t1 = oracle.intersect_box((2.0, 2.0, -1.0), (0.0, 0.0, 1.0))
    assert hit and t0 == 1.0 and t1 == 2.0
    # grazing along a face: the slab of a zero direction component is (+FLT_MAX, -FLT_MAX) -> min/max still give a hit
    hit, t0, t1 = oracle.intersect_box((0.0, 0.5, -1.0), (0.0, 0.0, 1.0))
    assert hit and t0 == 1.0 and t1 == 2.0
    # pointing away
    hit, _, _ = oracle.intersect_box((0.5, 0.5, -1.0), (0.0, 0.0, -1.0))
    assert not hit
    # diagonal
    hit, t0, t1 = oracle.intersect_box((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0))
    assert hit and t0 == 1.0 and t1 == 2.0 and t1 < FMAX


def _ramp_scene(ovr, oracle, n=4, dtype=np.float32, convention=0):
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    vol = (x + 10 * y + 100 * z).astype(dtype)
    colors, alphas, vr = ovr.synth.make_tfn("dense", 8, np.float32)
    cam = ovr.synth.make_camera("front", n)
    return oracle.OracleScene(vol, colors, alphas, (0.0, 333.0), cam, 8, 8, convention=convention), vol


def test_trilinear_voxel_centres_and_corners(ovr, oracle):
    sc, vol = _ramp_scene(ovr, oracle)
    n = 4
    for (i, j, k) in [(0, 0, 0), (1, 2, 3), (3, 3, 3), (2, 0, 1)]:
        p = ((i + .5) / n, (j + .5) / n, (k + .5) / n)   # cell-centred: texel centres
        assert sc.sample(p) == vol[k, j, i]
    # half-way between texels 1 and 2 along x: exact mean of a linear ramp
    assert sc.sample((2.0 / n, 0.5 / n, 0.5 / n)) == pytest.approx(1.5, abs=1e-6)
    # clamp-to-edge: outside [0,1] and inside the first/last half texel the edge value is returned
    assert sc.sample((-3.0, 0.5 / n, 0.5 / n)) == vol[0, 0, 0]
    assert sc.sample((0.01, 0.5 / n, 0.5 / n)) == vol[0, 0, 0]
    assert sc.sample((7.0, 7.0, 7.0)) == vol[3, 3, 3]
    # vertex-centred convention: texel k at k/(N-1)
    sv, _ = _ramp_scene(ovr, oracle, convention=1)
    assert sv.sample((1.0 / 3, 2.0 / 3, 1.0)) == pytest.approx(float(vol[3, 2, 1]), abs=1e-4)
    assert sv.sample((0.5, 0.0, 0.0)) == pytest.approx(1.5, abs=1e-5)


def test_gradient_forward_difference_and_flip(ovr, oracle):
    sc, vol = _ramp_scene(ovr, oracle)
    n = 4
    p = (1.5 / n, 1.5 / n, 1.5 / n)
    v = sc.sample(p)
    g = sc.gradient(p, v)
    # ramp x + 10 y + 100 z, one voxel = 1/n in object space -> gradient (1, 10, 100) * n
    assert np.allclose(g, [1 * n, 10 * n, 100 * n], rtol=1e-5)
    # at the upper bound the difference is taken backwards (shaders_common.h:204-210): same slope
    # central tap sits in the last half texel (value clamps to texel 3 = 300), backward tap at texel coordinate 2.1 = 210
    p = (1.5 / n, 1.5 / n, 1.0 - 0.4 / n)
    g = sc.gradient(p, sc.sample(p))
    assert g[2] == pytest.approx((210.0 - 300.0) / (-1.0 / n), rel=1e-4)


def test_integer_normalize(oracle):
    lib = oracle.load()
    f = lib.ovr_oracle_integer_normalize
    assert f(255.0, 100) == 1.0 and f(0.0, 100) == 0.0
    assert f(127.0, 101) == 1.0 and f(-128.0, 101) == -1.0
    assert f(65535.0, 200) == 1.0
    assert f(-32768.0, 201) == -1.0
    assert f(0.25, 400) == 0.25


def test_tfn_nodal_lerp(ovr, oracle):
    n = 5
    colors = np.zeros((n, 3), np.float32)
    colors[:, 0] = [0.0, 0.25, 0.5, 0.75, 1.0]
    alphas = np.stack([np.linspace(0, 1, n), [0.0, 0.1, 0.4, 0.9, 1.0]], 1).astype(np.float32)
    vol = np.zeros((2, 2, 2), np.float32)
    sc = oracle.OracleScene(vol, colors, alphas, (0.0, 4.0), ovr.synth.make_camera("front", 2), 4, 4)
    assert sc.tfn(0.0)[3] == 0.0
    assert sc.tfn(4.0)[3] == 1.0
    assert sc.tfn(1.0)[3] == pytest.approx(0.1)            # exactly on node 1
    assert sc.tfn(1.5)[3] == pytest.approx(0.25)           # half way between nodes 1 and 2
    assert sc.tfn(0.5 / (n - 1) * 4.0)[0] == pytest.approx(0.125 / 1.0 * 0.25 / 0.25 * 0.125 / 0.125, abs=1e-6)
    assert sc.tfn(-7.0)[3] == 0.0 and sc.tfn(99.0)[3] == 1.0  # clamped to the value range


def test_powf_is_restated_as_cuda_defines_the_intrinsic(oracle):
    """__powf(x, y) = exp2f(y * __log2f(x)) (CUDA C programming guide, intrinsic functions; shaders_raymarching.cu:64-66,118-122): the default
    opacity correction is 1 - exp2f(float(dt * log2f(1 - a))) evaluated in float32 steps, libm's powf is the switch"""
    f = oracle.load().ovr_oracle_opacity_correction
    assert oracle.load().ovr_oracle_get_powf_mode() == oracle.POWF_EXP2_LOG2
    rng = np.random.default_rng(11)
    a = np.concatenate([rng.random(2000), 10.0 ** rng.uniform(-8, -1, 2000), 1.0 - 10.0 ** rng.uniform(-7, -1, 2000)]).astype(np.float32)
    dt = np.concatenate([rng.uniform(0.01, 2.0, 3000), np.full(3000, 0.25)]).astype(np.float32)
    got = np.array([f(float(x), 1.0, float(t)) for x, t in zip(a, dt)], dtype=np.float32)
    x = (np.float32(1.0) - a).astype(np.float32)
    l = np.log2(x.astype(np.float64))                      # log2f is correctly rounded to < 1 ulp: compare through float64 with 1-ulp slack on each step
    m = (dt.astype(np.float64) * l.astype(np.float32).astype(np.float64)).astype(np.float32)
    want = np.clip(np.float32(1.0) - np.exp2(m.astype(np.float64)).astype(np.float32), 0.0, 1.0)
    keep = np.abs(dt - 1.0) >= 1e-7
    assert np.max(np.abs(got[keep] - want[keep])) <= 2.4e-7           # two float steps next to 1 (log2f / exp2f within an ulp of the float64 route)
    # the two modes differ exactly where rounds 1-4's tolerances came from: on the 6e-8 grid of 1 - x next to 1
    old = oracle.set_powf_mode(oracle.POWF_LIBM)
    libm = np.array([f(float(x), 1.0, float(t)) for x, t in zip(a, dt)], dtype=np.float32)
    oracle.set_powf_mode(old)
    assert np.max(np.abs(libm - got)) <= 2.4e-7 and np.any(libm != got)
    small = (a < 1e-6) & keep
    assert np.all(np.abs(libm[small] - got[small]) <= 1.2e-7)
    for m_ in (oracle.POWF_EXP2_LOG2, oracle.POWF_LIBM):
        o = oracle.set_powf_mode(m_)
        assert f(1.0, 1.0, 0.5) == 1.0 and f(0.0, 1.0, 0.5) == 0.0 and f(0.3, 1.0, 1.0) == np.float32(0.3)
        assert f(1.5, 1.0, 0.5) == 0.0                       # 1 - a < 0: NaN through log2 / pow, clamp01(NaN) = 0 (fminf / fmaxf, gdt.h:118-120)
        oracle.set_powf_mode(o)


def test_opacity_correction(oracle):
    f = oracle.load().ovr_oracle_opacity_correction
    assert f(0.3, 1.0, 1.0) == pytest.approx(0.3)                       # |base*dt - 1| < 1e-7 -> untouched
    assert f(0.3, 1.0, 0.25) == pytest.approx(1 - 0.7 ** 0.25, rel=1e-6)
    assert f(0.3, 1.0, 10.0) == pytest.approx(1 - 0.7 ** 10, rel=1e-6)
    assert f(1.0, 1.0, 0.5) == 1.0 and f(0.0, 1.0, 0.5) == 0.0


def test_exp_det_accuracy(oracle):
    f = oracle.load().ovr_oracle_exp_det
    xs = np.linspace(-30.0, 0.0, 2001, dtype=np.float32)
    got = np.array([f(float(x)) for x in xs])
    assert np.max(np.abs(got - np.exp(xs.astype(np.float64))) / np.exp(xs.astype(np.float64))) < 3e-6


def test_sparse_mask_against_numpy(oracle):
    rng = np.random.default_rng(5)
    noise = (rng.integers(0, 256, size=(64, 64, 64)) / 255.0).astype(np.float32)   # layout [y][x][t], values k/255
    w, h, frame = 40, 24, 7
    centre, scale, base = (0.4, 0.6), 0.3, 0.1
    got = oracle.sparse_mask(frame, w, h, centre, scale, base, noise).reshape(-1, 2)
    exp = []
    f = oracle.load().ovr_oracle_exp_det
    for y in range(h):
        for x in range(w):
            fx = np.float32(x) / np.float32(w) - np.float32(centre[0])
            fy = (np.float32(y) / np.float32(h) - np.float32(centre[1])) / (np.float32(w) / np.float32(h))
            arg = np.float32(-0.5) * (fx * fx + fy * fy) * (np.float32(1.0) / (np.float32(scale) * np.float32(scale)))
            p = (np.float32(1.0) - np.float32(base)) * np.float32(f(float(arg))) + np.float32(base)
            if noise[y % 64, x % 64, frame % 64] < p:
                exp.append((x, y))
    assert np.array_equal(got, np.array(exp, dtype=np.int32))
    assert 0 < len(exp) < w * h


def test_rgba8_quantisation(oracle):
    img = np.zeros((2, 3, 4), np.float32)
    img[0, 0] = [0.0, 1.0, 0.5, 2.0]
    img[1, 2] = [-1.0, 0.999, 1.0 / 255.0, 254.9 / 255.0]
    out = oracle.rgba8(img, flip=True)
    assert list(out[1, 0]) == [0, 255, 127, 255]            # row 0 of the frame is the LAST row of the PNG
    assert list(out[0, 2]) == [0, 254, 1, 254]


@pytest.fixture(params=["exp2_log2", "libm"])
def powf_golden(request, oracle):
    """both restatements of __powf (shaders_raymarching.cu:64-66,118-122) with their golden file: the default exp2f(y * log2f(x)) - CUDA's
    definition of the intrinsic, round 5 - and libm's powf, whose file is the frames.npz rounds 1-4 committed (the switch reproduces the old
    oracle bit for bit)"""
    mode, name = (oracle.POWF_EXP2_LOG2, "frames.npz") if request.param == "exp2_log2" else (oracle.POWF_LIBM, "frames_powf_libm.npz")
    old = oracle.set_powf_mode(mode)
    yield np.load(os.path.join(HERE, "golden", name))
    oracle.set_powf_mode(old)


@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_frames(ovr, oracle, name, powf_golden):
    gold = powf_golden
    case = make_case(ovr, oracle, **CASES[name])
    rgba, grad, cnt = oracle_scene(oracle, case).render(nthreads=2)
    assert np.array_equal(rgba, gold[name + "/rgba"])
    assert np.array_equal(grad, gold[name + "/grad"])
    assert [cnt.rays, cnt.samples, cnt.shaded_samples, cnt.shadow_samples, cnt.shadow_samples_visible] == list(gold[name + "/counters"])
    assert cnt.samples > 0 and np.isfinite(rgba).all()


def test_golden_accumulation(ovr, oracle, powf_golden):
    gold = powf_golden
    case = make_case(ovr, oracle, n=12, tf="sparse", cam="oblique", size=(24, 16), shading=2, spp=2)
    rgba, _, cnt = oracle_scene(oracle, case).render(frames=3, accumulate=True, nthreads=2)
    assert np.array_equal(rgba, gold["accum3_spp2/rgba"])
    assert cnt.rays == 24 * 16 * 2


def test_shading_modes_are_nested(ovr, oracle):
    """alpha never depends on shading; the shadow term can only darken"""
    case = make_case(ovr, oracle, n=16, tf="bumps", cam="oblique", size=(24, 16), shading=0)
    a0 = oracle_scene(oracle, case).render()[0]
    case["shading"] = 1
    a1 = oracle_scene(oracle, case).render()[0]
    case["shading"] = 2
    a2 = oracle_scene(oracle, case).render()[0]
    assert np.array_equal(a0[..., 3], a1[..., 3]) and np.array_equal(a1[..., 3], a2[..., 3])
    assert (a2[..., :3] <= a1[..., :3] + 1e-6).all()


# ---- round 2: data-range fallback, blue-noise jitter, zero-opacity shortcut, thread pool, EXR half conversion ----------

@pytest.mark.parametrize("dtype", [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32, np.float64])
def test_data_range_is_the_normalized_min_max(ovr, oracle, dtype):
    """compute_scalar_range + cuda_scalar_range (array.cpp:27-66,92-108): integer types of 8 and 32 bits are normalized,
    u16 / i16 / f64 are converted to float first (array.cpp:335-345) and keep raw values"""
    case = make_case(ovr, oracle, n=12, dtype=dtype)
    vol = case["vol"]
    sc = oracle_scene(oracle, case)
    lo, hi = sc.data_range()
    vmin, vmax = vol.min(), vol.max()
    f = np.float32
    exp = {np.uint8: (f(vmin) / f(255), f(vmax) / f(255)),
           np.int8: (max(f(vmin) / f(127), f(-1)), max(f(vmax) / f(127), f(-1))),
           np.uint32: (f(vmin) / f(4294967295), f(vmax) / f(4294967295)),
           np.int32: (max(f(vmin) / f(2147483647), f(-1)), max(f(vmax) / f(2147483647), f(-1)))}.get(dtype, (f(vmin), f(vmax)))
    assert (f(lo), f(hi)) == (f(exp[0]), f(exp[1]))


def test_invalid_tf_range_falls_back_to_the_data_range(ovr, oracle):
    """volume.cpp:135-142: with hi < lo (the default (1, -1)) the range found at load time stays in effect"""
    case = make_case(ovr, oracle, n=16, size=(24, 16))
    sc = oracle_scene(oracle, case)
    lo, hi = sc.data_range()
    explicit = dict(case, vr=(lo, hi))
    fallback = dict(case, vr=(1.0, -1.0))
    a, _, ca = oracle_scene(oracle, explicit).render()
    b, _, cb = oracle_scene(oracle, fallback).render()
    assert np.array_equal(a, b) and ca.samples == cb.samples and ca.shaded_samples == cb.shaded_samples
    other, _, _ = oracle_scene(oracle, dict(case, vr=(0.0, 1.0))).render()
    assert not np.array_equal(a, other)   # the synthetic field does not span [0, 1] exactly


def test_zero_opacity_shortcut_is_bit_identical(ovr, oracle):
    """the work the GPU skips (gradient taps + shadow march of samples with opacity exactly 0) does not change a bit"""
    for tf, shading in (("sparse", 2), ("bumps", 2), ("sparse", 1)):
        case = make_case(ovr, oracle, n=24, tf=tf, cam="oblique", size=(40, 28), shading=shading)
        plain, gp, cp = oracle_scene(oracle, case).render()
        fast, gf, cf = oracle_scene(oracle, case, skip_zero_opacity=True).render()
        assert np.array_equal(plain, fast) and np.array_equal(gp, gf)
        assert (cp.samples, cp.shaded_samples) == (cf.samples, cf.shaded_samples)
        if shading == 2:
            assert cf.shadow_samples == cp.shadow_samples_visible < cp.shadow_samples


def test_thread_pool_gives_the_same_frame_for_any_thread_count(ovr, oracle):
    case = make_case(ovr, oracle, n=16, cam="oblique", size=(37, 23), spp=2)
    ref, gr, cr = oracle_scene(oracle, case).render(frames=2, accumulate=True, nthreads=1)
    for nt in (2, 5, 3, 16):   # the pool is resized between calls
        a, g, c = oracle_scene(oracle, case).render(frames=2, accumulate=True, nthreads=nt)
        assert np.array_equal(a, ref) and np.array_equal(g, gr)
        assert (c.rays, c.samples, c.shaded_samples, c.shadow_samples) == (cr.rays, cr.samples, cr.shaded_samples, cr.shadow_samples)


def test_blue_noise_jitter_lookup_and_effect(ovr, oracle):
    import ctypes as C
    noise = ovr.synth.make_noise_tile(16, seed=3)
    case = make_case(ovr, oracle, n=16, cam="oblique", size=(40, 24), spp=2)
    sc = oracle_scene(oracle, case, jitter=1, noise=noise)
    out = (C.c_float * 2)()
    for (ix, iy, frame, k) in [(0, 0, 1, 0), (17, 5, 1, 1), (39, 23, 3, 0), (8, 8, 40, 1)]:
        sc.lib.ovr_oracle_jitter(C.byref(sc.s), ix, iy, frame, k, out)
        t = ((frame - 1) * 2 + k) % 64
        assert out[0] == noise[iy % 16, ix % 16, t] and out[1] == noise[(iy + 8) % 16, (ix + 8) % 16, t]
    # jitter applies even with one sample per pixel, and changes from frame to frame
    case1 = dict(case, spp=1)
    plain, _, _ = oracle_scene(oracle, case1).render()
    j1, _, _ = oracle_scene(oracle, case1, jitter=1, noise=noise).render()
    assert not np.array_equal(plain, j1)
    acc2, _, _ = oracle_scene(oracle, case1, jitter=1, noise=noise).render(frames=2, accumulate=True)
    assert not np.array_equal(acc2, j1) and np.isfinite(acc2).all()


def test_exr_half_conversion_rule(oracle):
    """tinyexr's float_to_half_full (the reference's EXR pixels): nearest with ties AWAY from zero, float denormals -> 0,
    carry into the exponent, overflow -> inf, NaN -> 0x7e00"""
    lib = oracle.load()
    h = lambda x: lib.ovr_oracle_float_to_half(float(np.float32(x)))
    assert h(0.0) == 0 and h(-0.0) == 0x8000 and h(1.0) == 0x3c00 and h(-2.0) == 0xc000
    assert h(65504.0) == 0x7bff and h(65520.0) == 0x7c00 and h(1e9) == 0x7c00   # 65520 is the tie to infinity: rounds up
    assert h(float("inf")) == 0x7c00 and h(float("-inf")) == 0xfc00 and h(float("nan")) & 0x7fff == 0x7e00
    assert h(1.0 + 2.0 ** -11) == 0x3c01          # exact tie between 0x3c00 and 0x3c01: away from zero (RNE would give 0x3c00)
    assert h(1.0 + 3 * 2.0 ** -11) == 0x3c02      # tie between 0x3c01 and 0x3c02: up (RNE agrees here)
    assert h(2.0 ** -24) == 1 and h(2.0 ** -25) == 1 and h(2.0 ** -26) == 0   # subnormal halves; half the smallest rounds up
    assert h(1e-45) == 0                           # float denormal -> 0
    assert h(2.0 - 2.0 ** -12) == 0x4000          # mantissa carry into the exponent
    # round trip through half -> float is exact for every finite half
    for bits in list(range(0, 0x7c00, 97)) + [0x7bff, 0x0001, 0x03ff, 0x0400]:
        f = lib.ovr_oracle_half_to_float(bits)
        assert lib.ovr_oracle_float_to_half(f) == bits
