"""What is left between the HIP kernels and the CPU oracle once the kernels' four DELIBERATE departures from the oracle's float arithmetic are taken back:
NOTHING - not a bit of any frame, not a count.

The product (libovr_hip.so) differs from the oracle in four named places: the opacity correction's `__powf`
(shaders_raymarching.cu:64-66,118-122) is v_exp_f32(y * v_log_f32(x)) where the oracle calls libm's exp2f / log2f (and the reference ex2.approx / lg2.approx:
three different last bits on the 6e-8 grid next to 1 that 1 - (1 - a)^dt lives on); the three per-sample normalisations multiply with v_rsq_f32; the gradient
multiplies with a reciprocal; 8-bit voxels are normalised once behind the filter instead of one by one in front of it.  libovr_hip_parity.so is THE SAME kernel
source compiled with -DOVR_PARITY_EXACT=1 (a test instrument: never loaded by default, never the product): those four take the oracle's form, the pow being a
machine-independent log2 / exp2 pair that the oracle evaluates too (mode "det").  With it, on the GPU: the 21 shipped scenes' frames (RGBA and gradient layer),
C1's full frame and the hunt sweep's 450 configurations (every voxel type, layout, pipeline, skipping, sparse sampling, shards, jitter, accumulation) equal the
oracle's BIT FOR BIT and every counter - primary, shaded, shadow - exactly.  So the product's tolerated differences (shaded counts within a scene's borderline
samples, one shadow iteration in 5.67 M at C1, one primary sample in 2 016 hunt cases, <= 1 on 8 bits) come from those four approximations and from nothing
else: the kernels' operation order IS the oracle's.  Round 5, VERDICT r4 #2."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DETLIB = os.path.join(ROOT, "open-volume-renderer_amd", "libovr_hip_parity.so")


def test_the_product_library_knows_the_deterministic_pair_too(ovr, oracle, hip_renderer_factory):
    """the known-answer entry: which = 1 evaluates the deterministic pair in the PRODUCT library - equal to the oracle's bit for bit; which = 0 is the
    product's own pow, v_exp_f32(y * v_log_f32(x)): within 2 ulp-of-1 steps of it next to 1, and not the same function"""
    assert ovr._lib.load().ovr_hip_built_for_exact_parity() == 0      # the library the tests load by default is the product
    rng = np.random.default_rng(3)
    a = np.concatenate([rng.random(20000), 10.0 ** rng.uniform(-8, -1, 20000)]).astype(np.float32)
    x, y = (np.float32(1.0) - a).astype(np.float32), np.concatenate([rng.uniform(0.01, 12.0, 20000), np.full(20000, 0.25)]).astype(np.float32)
    lib = oracle.load()
    want = np.array([lib.ovr_oracle_det_powf(float(p), float(e)) for p, e in zip(x, y)], dtype=np.float32)
    ren = hip_renderer_factory()
    det, hw = ren.pow_floats(x, y, 1), ren.pow_floats(x, y, 0)
    assert np.array_equal(det.view(np.uint32), want.view(np.uint32))
    # (relative: a product y * log2(x) near -100 carries its own rounding, 7.6e-6, into the exponent)
    assert np.max(np.abs(hw - want) / np.maximum(want, 1e-30)) < 3e-5 and np.any(hw != want)
    near1 = want > 0.5
    assert np.max(np.abs(hw - want)[near1]) <= 2.4e-7   # next to 1, where the opacity correction lives: within two 6e-8 steps of each other


@pytest.mark.parametrize("parts", [["kat", "scenes", "c1"], ["sweep"]], ids=["scenes_and_c1", "hunt_sweep_seed_303"])
def test_counts_are_exact_with_the_same_pow_on_both_sides(parts):
    assert os.path.exists(DETLIB), "libovr_hip_parity.so is missing: make -C open-volume-renderer_amd/csrc parity (build() does)"
    env = dict(os.environ, OVR_HIP_LIBRARY=DETLIB, OVR_ORACLE_POWF="det", OVR_DETPOW_CASES=os.environ.get("OVR_DETPOW_CASES", "450"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "parity_exact_check.py")] + parts, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "all exact" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_the_parity_suites_pass_with_equality_instead_of_tolerance():
    """The product's parity tests - tests/helpers.py::compare: <= 1 on 8 bits, 2e-4 in float - run once more against the exact-parity build with the bar replaced by
    EQUALITY of every float of every frame (OVR_PARITY_EXACT_RUN=1; the oracle in its "det" mode): the known-answer and frame tests, both configuration sweeps, the
    shipped scenes, the layouts / pipelines / skipping / sparse-sampling / shard / accumulation tests of rounds 2-4, the state-machine fuzzers and C1's full frame.
    (Left out: what tests the plugin boundary with the product library, bench.py, and the full-size invariants that involve no oracle.)"""
    assert os.path.exists(DETLIB), "libovr_hip_parity.so is missing: make -C open-volume-renderer_amd/csrc parity (build() does)"
    env = dict(os.environ, OVR_HIP_LIBRARY=DETLIB, OVR_PARITY_EXACT_RUN="1", OVR_ORACLE_POWF="det")   # (the variable: for the scripts the suites start themselves)
    t = os.path.join(ROOT, "tests")
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(t, "test_parity_gpu.py"), os.path.join(t, "test_config_sweep_gpu.py"), os.path.join(t, "test_shipped_scenes_gpu.py"), os.path.join(t, "test_robustness_gpu.py"),
           os.path.join(t, "test_round2_gpu.py"), os.path.join(t, "test_round3_gpu.py"), os.path.join(t, "test_round4_gpu.py"), os.path.join(t, "test_fuzz_states_gpu.py"),
           os.path.join(t, "test_full_size_gpu.py") + "::test_c1_full_frame_vs_oracle",
           "-k", "not bench and not stand_in and not two_ranks and not plugin and not renderbatch"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = out.stdout[-2500:] + out.stderr[-1500:]
    assert out.returncode == 0, tail
    import re
    m = re.search(r"(\d+) passed", out.stdout)
    assert m and int(m.group(1)) >= 300 and "failed" not in out.stdout.splitlines()[-1], tail
