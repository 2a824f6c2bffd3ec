"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from helpers import compare, hip_frame, hip_setup, make_case, oracle_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shading", [0, 1, 2])
@pytest.mark.parametrize("cam", ["front", "oblique"])
@pytest.mark.parametrize("tf", ["sparse", "dense", "bumps"])
def test_frame_parity_f32(ovr, oracle, hip_renderer_factory, shading, cam, tf):
    case = make_case(ovr, oracle, n=32, tf=tf, cam=cam, size=(64, 48), shading=shading)
    ref_rgba, ref_grad, cnt = oracle_scene(oracle, case).render()
    ren = hip_setup(ovr, hip_renderer_factory(), case)
    ren.render()
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
