"""Differential pin of the scene reader beyond the 21 shipped files: seeded mutations of the shipped scene JSONs - voxel types the shipped
scenes never use, `scales`, big endian, 1 ... 30 colour controls (unsorted, out-of-range positions), alpha arrays of any
resolution or none, 0 ... 5 gaussian objects (narrow, wide, tall), opacity control points, normalised and unnormalised mapping ranges,
sampling distances, cameras, additional lights - run through the REAL reference loader (oracle/_ref/ref_scene_probe, built by
oracle/build_ref.sh from the reference's sources) and stored with what it produced: tests/golden/scenes_fuzz/*.json (inputs: data) +
tests/golden/scenes_fuzz_expected.npz.  tests/test_scene_ingest.py compares this repo's reader with them.
Run where the reference tree is present:   python tests/golden/make_scene_fuzz_golden.py [count] [seed]     (--hunt N SEED: compare N
mutations with the reader right away, store nothing)"""
import base64
import copy
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SCENES = os.path.join(ROOT, "tests", "golden", "scenes")
PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_scene_probe")
TYPES = ["UNSIGNED_BYTE", "BYTE", "UNSIGNED_SHORT", "SHORT", "UNSIGNED_INT", "INT", "FLOAT", "DOUBLE"]


def _load(path):
    sys.path.insert(0, ROOT)
    import ovr_amd as ovr
    with open(path) as f:
        return json.loads(ovr.vidi3d._strip_json_comments(f.read()))


def mutate(doc, rng):
    d = copy.deepcopy(doc)
    ds, view = d["dataSource"][0], d["view"]
    jv = view["volume"]
    tf = jv["transferFunction"]
    f32 = lambda x: float(np.float32(x))
    if rng.integers(2):
        ds["type"] = str(rng.choice(TYPES))
    if rng.integers(3) == 0:
        ds["endian"] = str(rng.choice(["BIG_ENDIAN", "LITTLE_ENDIAN"]))
    if rng.integers(3) == 0:
        ds["scales"] = {k: f32(rng.choice([0.25, 0.5, 1.0, 1.5, 3.0])) for k in "xyz"}
    # colour controls
    if rng.integers(4) != 0:
        n = int(rng.integers(1, 31))
        pos = rng.uniform(-0.1, 1.1, n) if rng.integers(4) == 0 else rng.uniform(0.0, 1.0, n)
        # (no duplicate positions: the reference orders its controls with std::sort, which is not stable - which of two controls at one
        # position comes first is up to the library's sort, not to the loader)
        if rng.integers(2):
            pos = np.sort(pos)
        tf["colorControls"] = [{"color": {k: f32(rng.uniform(0, 1)) for k in "rgb"}, "position": f32(p)} for p in pos]
    # alpha array
    r = int(rng.integers(5))
    if r == 0:
        tf.pop("alphaArray", None)
        tf["resolution"] = int(rng.choice([16, 100, 256, 1024]))
    elif r <= 2:
        n = int(rng.choice([8, 64, 333, 1024, 2048]))
        kind = int(rng.integers(3))
        a = rng.uniform(0, 1, n) if kind == 0 else np.clip(np.linspace(-0.2, 1.2, n), 0, 1) if kind == 1 else (rng.uniform(0, 1, n) < 0.1) * rng.uniform(0, 1, n)
        if rng.integers(2):
            a[0] = rng.uniform(0, 0.02); a[-1] = rng.uniform(0, 0.02)        # around the "< 0.01 -> 0" rule at both ends
        tf["alphaArray"] = {"data": base64.b64encode(a.astype("<f4").tobytes()).decode(), "encoding": "BASE64"}
        tf["resolution"] = n if rng.integers(2) else int(rng.choice([64, 1024]))
    # gaussians
    if rng.integers(2):
        tf["gaussianObjects"] = [{"alphaArray": {"data": "", "encoding": "BASE64"}, "heightFactor": f32(rng.uniform(0.001, 0.5)), "mean": f32(rng.uniform(-0.1, 1.1)),
                                  "resolution": 1024, "sigma": f32(10 ** rng.uniform(-3, -0.3))} for _ in range(int(rng.integers(0, 6)))]
    if rng.integers(3) == 0:
        pts = sorted(rng.uniform(0, 1, int(rng.integers(1, 7)))) if rng.integers(2) else list(rng.uniform(0, 1, int(rng.integers(1, 7))))
        tf["opacityControl"] = [{"position": {"x": f32(p), "y": f32(rng.uniform(0, 1))}} for p in pts]
    # mapping range
    lo, hi = sorted(rng.uniform(0, 1, 2))
    r = int(rng.integers(3))
    if r == 0:
        jv["scalarMappingRange"] = {"minimum": float(lo), "maximum": float(hi)}
        jv.pop("scalarMappingRangeUnnormalized", None)
    elif r == 1:
        jv["scalarMappingRangeUnnormalized"] = {"minimum": float(lo * 1000 - 50), "maximum": float(hi * 1000 + 50)}
    jv["sampleDistance"] = float(rng.choice([0.05, 0.1, 0.25, 0.5, 1.0, 2.0, 0.3]))
    cam = view["camera"]
    if rng.integers(2):
        for k in ("eye", "center"):
            cam[k] = {a: float(rng.normal(0, 200)) for a in "xyz"}
        cam["up"] = {"x": float(rng.normal()), "y": float(rng.normal()), "z": float(rng.normal())}
        cam["fovy"] = float(rng.uniform(10, 100))
    if rng.integers(3) == 0:
        view["additionalLightSources"] = [dict(copy.deepcopy(view["lightSource"]), position={"w": 0, "x": float(rng.normal()), "y": float(rng.normal()), "z": float(rng.normal())},
                                               diffuse={"a": 1, "r": f32(rng.uniform(0, 1)), "g": f32(rng.uniform(0, 1)), "b": f32(rng.uniform(0, 1))}) for _ in range(int(rng.integers(1, 4)))]
    return d


def probe(paths):
    tmp = "/tmp/ovr_scene_fuzz_probe_out.json"
    subprocess.check_call([PROBE, tmp] + paths, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return json.load(open(tmp))


def compare(ovr, path, exp):
    """this repo's reader against what the reference loader produced; returns a list of mismatches"""
    bad = []
    d = ovr.vidi3d.read_scene(path, load_volume=False)
    g = lambda k: np.asarray(exp[k], dtype=np.float32)
    if ovr.renderer._NP_TO_TYPE[d["dtype"]] != int(exp["value_type"]):
        bad.append(("value_type", ovr.renderer._NP_TO_TYPE[d["dtype"]], exp["value_type"]))
    if not (np.allclose(d["grid_spacing"], g("grid_spacing")) and np.allclose(d["grid_origin"], g("grid_origin"))):
        bad.append(("grid", d["grid_spacing"], exp["grid_spacing"]))
    ref_c, ref_o = g("tfn_color").reshape(-1, 4), g("tfn_opacity")
    if d["tfn_color"].shape != ref_c.shape or d["tfn_opacity"].shape != ref_o.shape:
        bad.append(("tfn shape", d["tfn_color"].shape, ref_c.shape))
    else:
        if np.abs(d["tfn_color"] - ref_c).max() > 2e-6:
            bad.append(("tfn_color", float(np.abs(d["tfn_color"] - ref_c).max()), int(np.abs(d["tfn_color"] - ref_c).max(axis=1).argmax())))
        if np.abs(d["tfn_opacity"] - ref_o).max() > 2e-6:
            bad.append(("tfn_opacity", float(np.abs(d["tfn_opacity"] - ref_o).max()), int(np.abs(d["tfn_opacity"] - ref_o).argmax())))
    if not np.allclose(d["value_range"], g("value_range"), rtol=1e-6):
        bad.append(("value_range", d["value_range"], exp["value_range"]))
    eye, at, up, fovy = d["camera"]
    if not np.allclose(list(eye) + list(at) + list(up) + [fovy], g("camera"), rtol=1e-6):
        bad.append(("camera", d["camera"], exp["camera"]))
    lights = np.array([list(a) + list(b) for a, b in d["lights"]], np.float32).ravel()
    if lights.shape != g("lights").shape or not np.allclose(lights, g("lights"), rtol=1e-6):
        bad.append(("lights", lights.tolist(), exp["lights"]))
    if not np.isclose(d["volume_sampling_rate"], float(exp["volume_sampling_rate"][0]), rtol=1e-6):
        bad.append(("rate", d["volume_sampling_rate"], exp["volume_sampling_rate"]))
    return bad


def main():
    hunt = "--hunt" in sys.argv
    args = [a for a in sys.argv[1:] if a != "--hunt"]
    count = int(args[0]) if args else 36
    seed = int(args[1]) if len(args) > 1 else 20261004
    rng = np.random.default_rng(seed)
    bases = sorted(f for f in os.listdir(SCENES) if f.startswith("scene_") and f.endswith(".json"))
    outdir = "/tmp/ovr_scene_fuzz" if hunt else os.path.join(ROOT, "tests", "golden", "scenes_fuzz")
    os.makedirs(outdir, exist_ok=True)
    paths = []
    for i in range(count):
        base = bases[int(rng.integers(len(bases)))]
        d = mutate(_load(os.path.join(SCENES, base)), rng)
        p = os.path.join(outdir, f"fuzz_{i:03d}_{base}")
        json.dump(d, open(p, "w"))
        paths.append(p)
    doc = {}
    for k in range(0, len(paths), 20):
        doc.update(probe(paths[k:k + 20]))
    if hunt:
        sys.path.insert(0, ROOT)
        import ovr_amd as ovr
        nbad = 0
        for p in paths:
            try:
                bad = compare(ovr, p, doc[os.path.basename(p)])
            except Exception as e:   # the reader refused what the reference loads
                bad = [("exception", repr(e))]
            if bad:
                nbad += 1
                print(os.path.basename(p), str(bad)[:400])
        print(f"{count} mutated scenes, {nbad} differ")
        return
    out = {}
    for name, d in doc.items():
        for k, v in d.items():
            out[f"{name}/{k}"] = np.asarray(v, dtype=np.float32) if k != "value_type" else np.asarray(v, dtype=np.int32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "scenes_fuzz_expected.npz"), **out)
    print("wrote", len(out), "arrays for", len(doc), "mutated scenes")


if __name__ == "__main__":
    main()
