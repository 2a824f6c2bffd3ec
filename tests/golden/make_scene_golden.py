"""Generates tests/golden/scenes_expected.npz from the REAL reference scene loader (oracle/_ref/ref_scene_probe, built by
oracle/build_ref.sh from oracle/ref_scene_probe.cpp against the reference's sources) for ALL scene files the reference
ships under data/configs (21; copied to tests/golden/scenes/ as input fixtures - they are data, not code): uint8 / uint16 /
float volumes, 3...28 colour controls, 0...4 gaussian objects, base64 alpha arrays, scalarMappingRange with and without the
unnormalized variant, sampleDistance 0.25 and 0.05.
Run from the repo root where the reference tree is present:  python tests/golden/make_scene_golden.py"""
import json
import os
import shutil
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("OVR_ROOT", "/root/reference")
SCENES = sorted(f for f in os.listdir(os.path.join(REF, "data", "configs")) if f.endswith(".json")) if os.path.isdir(REF) else []


def main():
    dst = os.path.join(ROOT, "tests", "golden", "scenes")
    os.makedirs(dst, exist_ok=True)
    paths = []
    for s in SCENES:
        shutil.copyfile(os.path.join(REF, "data", "configs", s), os.path.join(dst, s))
        os.chmod(os.path.join(dst, s), 0o644)
        paths.append(os.path.join(dst, s))
    tmp = "/tmp/ovr_scene_probe_out.json"
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "ref_scene_probe"), tmp] + paths, stdout=subprocess.DEVNULL)
    doc = json.load(open(tmp))
    out = {}
    for name, d in doc.items():
        for k, v in d.items():
            out[f"{name}/{k}"] = np.asarray(v, dtype=np.float32) if k != "value_type" else np.asarray(v, dtype=np.int32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "scenes_expected.npz"), **out)
    print("wrote", len(out), "arrays for", len(doc), "scenes")


if __name__ == "__main__":
    main()
