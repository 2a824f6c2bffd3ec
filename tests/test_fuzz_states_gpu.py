"""A short run of the state-machine fuzzer (tests/fuzz_states.py: random setter / commit / render / swap sequences on one renderer, every mapped
frame against the oracle) - 24 episodes of 10 transitions with a fixed seed; wider runs are one-off hunts (profiles/r03_notes.md section 16)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_state_machine_fuzz():
    out = subprocess.run([sys.executable, os.path.join(HERE, "fuzz_states.py"), "24", "7", "10"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "24 episodes, 0 failed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("env", [{"OVR_FUZZ_GROUP": "3"}, {"OVR_FUZZ_GROUP": "5", "OVR_FUZZ_LAZY": "1"}, {"OVR_FUZZ_LAZY": "1", "OVR_FUZZ_NONFINITE": "1"}])
def test_state_machine_fuzz_round4(env):
    """the same walk on an in-process device group (3 / 5 members on one card: every frame must be the oracle's whole frame, through tile-size
    changes, swaps, sparse sampling, new volumes) and with replicas built in the background (the default layouts mode), non-finite voxels included"""
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, os.path.join(HERE, "fuzz_states.py"), "12", "11", "10"], capture_output=True, text=True, timeout=900, env=e)
    assert out.returncode == 0 and "12 episodes, 0 failed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_sparse_mask_random_configurations():
    """tests/mask_hunt.py: 60 random frame sizes (up to 4K), noise tiles, focus windows, base-noise levels and frame indices - the device's compacted
    pixel list equals the oracle's bit for bit"""
    out = subprocess.run([sys.executable, os.path.join(HERE, "mask_hunt.py"), "60", "5"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "60 mask configurations, 0 differ" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_image_shards_random_configurations():
    """tests/shard_hunt.py: 40 random frame sizes, tile shapes (1 x 1 ... 64 x 64, also wider than the frame) and world sizes 1 ... 8 - every rank's payload equals
    the host restatement and the scattered frame equals the unsharded one bit for bit"""
    out = subprocess.run([sys.executable, os.path.join(HERE, "shard_hunt.py"), "40", "5"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "40 shard configurations, 0 differ" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
