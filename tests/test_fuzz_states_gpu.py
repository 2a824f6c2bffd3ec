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
