"""Generates tests/golden/ref_probe_wide.npz with oracle/_ref/ref_probe_wide (oracle/ref_probe_wide.cpp, built by oracle/build_ref.sh against the
reference's sources): the REAL reference's image_to_rgba8, gdt camera / affine math and EXR float -> half round trip on wide seeded inputs.
Run where the reference tree is present:   python tests/golden/make_ref_probe_wide.py"""
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "ref_probe_wide"), d])
        out = {}
        for name in sorted(os.listdir(d)):
            stem, ext = name.rsplit(".", 1)
            raw = np.fromfile(os.path.join(d, name), dtype={"f32": "<u4", "u8": np.uint8}[ext])   # floats kept as bit patterns: NaN-safe, exact
            out[stem] = raw
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_probe_wide.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
