import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.load()
    if os.environ.get("OVR_PARITY_EXACT_RUN") == "1":
        # the whole parity suite against the exact-parity build of the kernels (tests/test_parity_exact_gpu.py): same pow on both sides, frames must be EQUAL
        import ovr_amd
        assert ovr_amd._lib.load().ovr_hip_built_for_exact_parity() == 1, "OVR_PARITY_EXACT_RUN needs OVR_HIP_LIBRARY = libovr_hip_parity.so"
        O.set_powf_mode(O.POWF_DET)
    return O


@pytest.fixture(scope="session")
def ovr():
    import ovr_amd
    return ovr_amd


@pytest.fixture(scope="session")
def hip_renderer_factory(ovr):
    """creates DeviceHIP instances; fails loudly (no skip, no fallback) when the library or the GPU is missing"""
    made = []

    def make():
        r = ovr.create_renderer("hip")
        made.append(r)
        return r

    yield make
    for r in made:
        r.close()
