"""Extracts the NUMERIC LITERALS of the reference's integration loop from the cited lines of its source text into tests/golden/ref_literals.json.

The fixture is data - name, file, line, value - not source text.  It pins the constants of oracle/ovr_oracle.c (ovr_oracle_literals) against what the
reference's files say; it does not pin the arithmetic around them (that stays "parity unpinned": DESIGN.md section 3).  Reads /root/reference (or
$OVR_ROOT) as text only; run here, never on the GPU box:  python tests/golden/make_ref_literals.py"""
import json
import os
import re
import struct

R = os.environ.get("OVR_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
NUM = r"[-+]?(?:0x[0-9a-fA-F]+|\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|\d+(?:[eE][-+]?\d+)?)"

# name -> (file, line, regex with ONE group around the literal, kind)
SPEC = {
    "ert_primary": ("ovr/devices/optix7/shaders_raymarching.cu", 110, r"payload\.alpha\s*<\s*(" + NUM + r")f", "f32"),
    "ert_shadow": ("ovr/devices/optix7/shaders_raymarching.cu", 65, r"payload\.alpha\s*<\s*(" + NUM + r")f", "f32"),
    "shadow_step_scale": ("ovr/devices/optix7/shaders_raymarching.cu", 221, r"self\.step\s*\*\s*(" + NUM + r")f", "f32"),
    "midpoint": ("ovr/devices/optix7/shaders_raymarching.cu", 112, r"org\s*\+\s*(" + NUM + r")f\s*\*\s*\(t\.x\s*\+\s*t\.y\)", "f32"),
    "midpoint_shadow": ("ovr/devices/optix7/shaders_raymarching.cu", 67, r"org\s*\+\s*(" + NUM + r")f\s*\*\s*\(t\.x\s*\+\s*t\.y\)", "f32"),
    "nearly_equal_eps": ("ovr/devices/optix7/shaders_common.h", 322, r"epsilon\s*=\s*(" + NUM + r")f", "f32"),
    "light_x": ("ovr/devices/optix7/params.h", 79, r"light_directional_pos\{\s*(" + NUM + r")f", "f32"),
    "light_y": ("ovr/devices/optix7/params.h", 79, r"light_directional_pos\{\s*" + NUM + r"f\s*,\s*(" + NUM + r")f", "f32"),
    "light_z": ("ovr/devices/optix7/params.h", 79, r"light_directional_pos\{\s*" + NUM + r"f\s*,\s*" + NUM + r"f\s*,\s*(" + NUM + r")f", "f32"),
    "light_rgb": ("ovr/devices/optix7/shaders_raymarching.cu", 138, r"light_rgb\s*=\s*vec3f\((" + NUM + r")f\)", "f32"),
    "shade_ambient": ("ovr/devices/optix7/shaders_raymarching.cu", 157, r"\*=\s*(" + NUM + r")f\s*\+", "f32"),
    "shade_diffuse": ("ovr/devices/optix7/shaders_raymarching.cu", 157, r"\+\s*(" + NUM + r")f\s*\*\s*cosNL", "f32"),
    "tea_rounds": ("ovr/common/random/random.h", 184, r"tea<(\d+)>", "int"),
    "tea_delta": ("ovr/common/random/random.h", 158, r"sum\s*\+=\s*(" + NUM + r")", "int"),
    "tea_k0": ("ovr/common/random/random.h", 159, r"\(v1\s*<<\s*4\)\s*\+\s*(" + NUM + r")", "int"),
    "tea_k1": ("ovr/common/random/random.h", 159, r"\(v1\s*>>\s*5\)\s*\+\s*(" + NUM + r")", "int"),
    "tea_k2": ("ovr/common/random/random.h", 160, r"\(v0\s*<<\s*4\)\s*\+\s*(" + NUM + r")", "int"),
    "tea_k3": ("ovr/common/random/random.h", 160, r"\(v0\s*>>\s*5\)\s*\+\s*(" + NUM + r")", "int"),
    "tea_tofloat": ("ovr/common/random/random.h", 185, r"tofloat\s*=\s*(" + NUM + r")f", "f32"),
    "pixel_jitter_centre": ("ovr/devices/optix7/shaders_raymarching.cu", 355, r"get_floats\(\)\)\s*-\s*(" + NUM + r")f", "f32"),
    "screen_centre_x": ("ovr/devices/optix7/shaders_raymarching.cu", 361, r"screen\.x\s*-\s*(" + NUM + r")f", "f32"),
    "screen_centre_y": ("ovr/devices/optix7/shaders_raymarching.cu", 362, r"screen\.y\s*-\s*(" + NUM + r")f", "f32"),
}
# the two limits are spelled through std::numeric_limits<float>: the fixture records WHICH member the line names
LIMITS = {
    "float_large": ("ovr/common/math_def.h", 56, r"float_large\s*=\s*std::numeric_limits<float>::(\w+)\(\)"),
    "float_small": ("ovr/common/math_def.h", 57, r"float_small\s*=\s*std::numeric_limits<float>::(\w+)\(\)"),
}


def f32(x):
    return struct.unpack("<f", struct.pack("<f", float(x)))[0]


def line_of(path, n, cache={}):
    if path not in cache:
        with open(os.path.join(R, path), errors="replace") as f:
            cache[path] = f.read().split("\n")
    return cache[path][n - 1]


def main():
    out = {}
    for name, (path, n, rx, kind) in SPEC.items():
        m = re.search(rx, line_of(path, n))
        assert m, (name, path, n)
        tok = m.group(1)
        if kind == "int":
            v = int(tok, 0)
        else:
            v = f32(tok)  # an `f`-suffixed literal is the nearest float32
        out[name] = {"file": path, "line": n, "kind": kind, "value": v}
    for name, (path, n, rx) in LIMITS.items():
        m = re.search(rx, line_of(path, n))
        assert m, (name, path, n)
        v = {"max": 3.4028234663852886e38, "min": 1.1754943508222875e-38}[m.group(1)]
        out[name] = {"file": path, "line": n, "kind": "numeric_limits<float>::" + m.group(1), "value": v}
    with open(os.path.join(HERE, "ref_literals.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote %d literals" % len(out))


if __name__ == "__main__":
    main()
