"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from helpers import compare, hip_frame, hip_setup, make_case, oracle_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pipeline", [1, 2])
@pytest.mark.parametrize("shading", [0, 1, 2])
@pytest.mark.parametrize("cam", ["front", "oblique"])
@pytest.mark.parametrize("tf", ["sparse", "dense", "bumps"])
def test_frame_parity_f32(ovr, oracle, hip_renderer_factory, shading, cam, tf, pipeline):
    if shading == 0 and pipeline == 2:
        pytest.skip("no shading: there is nothing to pool")
    case = make_case(ovr, oracle, n=32, tf=tf, cam=cam, size=(64, 48), shading=shading)
    ref_rgba, ref_grad, cnt = oracle_scene(oracle, case).render()
    ren = hip_setup(ovr, hip_renderer_factory(), case, pipeline=pipeline)
    ren.render()
    assert ren.stats().pipeline == (pipeline if shading else 1)
    rgba, grad = hip_frame(ovr, ren)
    compare(oracle, rgba, ref_rgba, name=f"{tf}/{cam}/{shading}")
    if shading:
        assert np.abs(grad - ref_grad).max() <= 2e-3
    st = ren.stats()
    assert st.rays == cnt.rays
    assert st.samples == cnt.samples, "primary sample count differs from the oracle"
    assert st.shaded_samples == cnt.shaded_samples
    if shading == 2:
        assert st.shadow_samples == cnt.shadow_samples_visible


@pytest.mark.parametrize("shading", [1, 2])
def test_pipelines_bit_identical(ovr, oracle, hip_renderer_factory, shading):
    """in-place and pooled shading apply every pixel's contributions in the same order: the frames must be equal bit for bit"""
    case = make_case(ovr, oracle, n=48, tf="bumps", cam="oblique", size=(160, 96), shading=shading)
    frames = []
    for pipeline in (1, 2):
        ren = hip_setup(ovr, hip_renderer_factory(), case, pipeline=pipeline)
        ren.render()
        frames.append(hip_frame(ovr, ren) + (ren.stats(),))
    assert np.array_equal(frames[0][0], frames[1][0])
    assert np.array_equal(frames[0][1], frames[1][1])
    for k in ("rays", "samples", "shaded_samples", "shadow_samples"):
        assert getattr(frames[0][2], k) == getattr(frames[1][2], k)
    assert frames[1][2].pool_chunks > 0
