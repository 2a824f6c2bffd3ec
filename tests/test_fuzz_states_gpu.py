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
