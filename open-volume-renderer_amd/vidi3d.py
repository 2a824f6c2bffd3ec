"""Writer for the reference's VIDI3D scene JSON (the format `create_json_scene` reads, reference
ovr/serializer/serializer_vidi3d.cpp:203-408 and extern/tfn/core.h:710-790).  The reference ships scene files but no volume
data; this writes a scene + raw volume pair that the UNMODIFIED reference apps can load, so `renderbatch --device hip`
can be run end to end on synthetic data."""
import base64
import json
import os

import numpy as np

_TYPE_NAME = {np.dtype(np.uint8): "UNSIGNED_BYTE", np.dtype(np.int8): "BYTE", np.dtype(np.uint16): "UNSIGNED_SHORT",
              np.dtype(np.int16): "SHORT", np.dtype(np.uint32): "UNSIGNED_INT", np.dtype(np.int32): "INT",
              np.dtype(np.float32): "FLOAT", np.dtype(np.float64): "DOUBLE"}


def write_scene(directory, name, volume, color_controls, alpha_table, scalar_range_normalized, camera, fovy=45.0,
                sample_distance=1.0):
    """volume: (nz, ny, nx) array; color_controls: [(position, r, g, b)]; alpha_table: float32 opacities (its length is the
    TF resolution); scalar_range_normalized: (lo, hi) as a fraction of the type's maximum for integer data, raw for floats
    (scalarMappingRange, serializer_vidi3d.cpp:236-272); camera: (eye, center, up)."""
    os.makedirs(directory, exist_ok=True)
    volume = np.ascontiguousarray(volume)
    raw = os.path.join(directory, name + ".raw")
    volume.tofile(raw)
    nz, ny, nx = volume.shape
    eye, center, up = camera
    alpha_table = np.ascontiguousarray(alpha_table, dtype="<f4")
    xyz = lambda v: {"x": float(v[0]), "y": float(v[1]), "z": float(v[2])}
    doc = {
        "dataSource": [{
            "dimensions": {"x": nx, "y": ny, "z": nz}, "endian": "LITTLE_ENDIAN", "fileName": [os.path.basename(raw), raw],
            "fileUpperLeft": False, "format": "REGULAR_GRID_RAW_BINARY", "id": 1, "name": os.path.basename(raw), "offset": 0,
            "type": _TYPE_NAME[volume.dtype]}],
        "snapshot": [],
        "view": {
            "camera": {"center": xyz(center), "eye": xyz(eye), "fovy": float(fovy), "projectionMode": "PERSPECTIVE", "up": xyz(up),
                       "zFar": 10000.0, "zNear": 1.0},
            "lightSource": {"ambient": {"a": 1, "b": 1, "g": 1, "r": 1}, "diffuse": {"a": 1, "b": 1, "g": 1, "r": 1},
                            "position": {"w": 0, "x": 0, "y": 0, "z": 1}, "specular": {"a": 1, "b": 1, "g": 1, "r": 1},
                            "type": "DIRECTIONAL_LIGHT"},
            "volume": {
                "dataId": 1, "sampleDistance": float(sample_distance), "opacityUnitDistance": 1,
                "scalarMappingRange": {"minimum": float(scalar_range_normalized[0]), "maximum": float(scalar_range_normalized[1])},
                "transferFunction": {
                    "alphaArray": {"data": base64.b64encode(alpha_table.tobytes()).decode("ascii"), "encoding": "BASE64"},
                    "colorControls": [{"color": {"r": float(r), "g": float(g), "b": float(b)}, "position": float(p)}
                                      for (p, r, g, b) in color_controls],
                    "resolution": int(alpha_table.size)},
                "transferFunctionType": "TRANSFER_FUNCTION", "visible": True},
        },
    }
    path = os.path.join(directory, name + ".json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    return path


# ------------------------------------------------------------------------------------------------------------------------
# Scene ingestion: VIDI3D JSON -> Scene, a restatement of the reference's loader
#   create_json_scene_vidi3d ........ ovr/serializer/serializer_vidi3d.cpp:334-408
#   create_scene_tfn ................ :203-277       create_scene_volume :279-308      create_scene_camera :310-324
#   tfn::loadTransferFunction ....... extern/tfn/core.h:710-790   updateColorMap :598-634   GaussianObject :349-378
#   CreateArray3DScalarFromFile ..... ovr/scene.cpp:181-245 (raw file, byte offset, endian swap)
# Pinned against the real reference loader for six shipped scenes (tests/test_scene_ingest.py, tests/golden/scenes_expected.npz).
# ------------------------------------------------------------------------------------------------------------------------
_NAME_TYPE = {v: k for k, v in _TYPE_NAME.items()}
_F = np.float32


def _strip_json_comments(text):
    """the reference parses with nlohmann's ignore_comments = true (serializer_vidi3d.cpp:417)"""
    out, i, n, in_str = [], 0, len(text), False
    while i < n:
        c = text[i]
        if in_str:
            out.append(c)
            if c == "\\":
                out.append(text[i + 1]); i += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True; out.append(c)
        elif text.startswith("//", i):
            while i < n and text[i] != "\n":
                i += 1
            continue
        elif text.startswith("/*", i):
            i = text.index("*/", i) + 2
            continue
        else:
            out.append(c)
        i += 1
    return "".join(out)


def rasterize_transfer_function(jstfn):
    """tfn::loadTransferFunction + TransferFunctionCore::updateColorMap: returns the RGBA table (resolution x 4, float32)"""
    resolution = int(jstfn.get("resolution", 1024))
    alpha = None
    aa = jstfn.get("alphaArray")
    if aa and "data" in aa and aa.get("encoding") == "BASE64":
        raw = base64.b64decode(aa["data"])
        resolution = len(raw) // 4
        alpha = np.frombuffer(raw[: resolution * 4], dtype="<f4").astype(_F)
    if alpha is None:
        alpha = np.zeros(resolution, _F)
    controls = [(_F(c["position"]), [_F(c["color"][k]) for k in "rgb"]) for c in jstfn.get("colorControls", [])
                if "position" in c and "color" in c]
    if not controls:
        controls = [(_F(0), [_F(0)] * 3)]
    controls.sort(key=lambda c: c[0])                      # std::sort with operator< on position
    pos = np.array([c[0] for c in controls], _F)
    col = np.array([c[1] for c in controls], _F)
    table = np.zeros((resolution, 4), _F)
    gaussians = []
    for g in jstfn.get("gaussianObjects", []):
        if not all(k in g for k in ("mean", "sigma", "heightFactor")):
            continue
        mean, sigma, hf = _F(g["mean"]), _F(g["sigma"]), _F(g["heightFactor"])
        x = (np.arange(resolution, dtype=_F) + _F(0.5)) * (_F(1.0) / _F(resolution))
        diff = x - mean
        val = hf / (sigma * np.sqrt(_F(2.0) * _F(np.pi))) * np.exp(-(diff * diff) / (_F(2.0) * sigma * sigma))
        gaussians.append(np.clip(val.astype(_F), _F(0), _F(1)))
    ub = 0
    for i in range(resolution):
        value = (_F(i) + _F(0.5)) / _F(resolution)
        while ub < len(pos) and not (value < pos[ub]):     # std::upper_bound, resumed from the previous position
            ub += 1
        if ub <= 0:
            c = col[0]
        elif ub >= len(pos):
            c = col[ub - 1]
        else:
            w = np.abs(value - pos[ub - 1]) / np.abs(pos[ub] - pos[ub - 1])
            c = col[ub - 1] + w * (col[ub] - col[ub - 1])  # mix(x, y, a) = x + a*(y - x)
        a = alpha[i]
        for g in gaussians:
            a = max(a, g[i])
        table[i, :3] = c
        table[i, 3] = a
    ocs = [oc["position"] for oc in jstfn.get("opacityControl", []) if "position" in oc]
    if ocs:                                                  # updateFromAlphaControls, core.h:652-685
        pts = sorted(((float(p["x"]), float(p["y"])) for p in ocs), key=lambda p: _F(p[0]))
        px = np.array([p[0] for p in pts], _F)
        py = np.array([p[1] for p in pts], _F)
        ub = 0
        for i in range(resolution):
            value = float(i) / float(resolution - 1)
            while ub < len(px) and not (_F(value) < px[ub]):
                ub += 1
            if ub <= 0:
                a = float(py[0])
            elif ub >= len(px):
                a = float(py[ub - 1])
            else:
                w = abs(value - float(px[ub - 1])) / abs(float(px[ub]) - float(px[ub - 1]))
                a = float(py[ub - 1] + _F(w) * (py[ub] - py[ub - 1]))
            table[i, 3] = max(table[i, 3], _F(a))
    return table


_INT_MAX = {np.dtype(np.uint8): 255.0, np.dtype(np.int8): 127.0, np.dtype(np.uint16): 65535.0, np.dtype(np.int16): 32767.0,
            np.dtype(np.uint32): 4294967295.0, np.dtype(np.int32): 2147483647.0}


def read_scene(path, load_volume=True):
    """VIDI3D scene file -> dict(volume, dtype, dims, grid_origin, grid_spacing, tfn_color (N x 4), tfn_opacity (N), value_range,
    camera (eye, at, up, fovy), lights [(direction, color)], volume_sampling_rate).  `scene_from_file` wraps it into a Scene."""
    with open(path) as f:
        root = json.loads(_strip_json_comments(f.read()))
    workdir = os.path.dirname(os.path.abspath(path)) or "."
    ds = root["dataSource"][0]
    view = root["view"]
    jv = view["volume"]
    if ds["format"] != "REGULAR_GRID_RAW_BINARY":
        raise RuntimeError("data type unimplemented")                                   # serializer_vidi3d.cpp:304
    dtype = _NAME_TYPE[ds["type"]]
    nx, ny, nz = int(ds["dimensions"]["x"]), int(ds["dimensions"]["y"]), int(ds["dimensions"]["z"])
    volume = None
    if load_volume:
        names = ds["fileName"] if isinstance(ds["fileName"], list) else [ds["fileName"]]
        fn = None
        for cand in names:                                                              # valid_filename, :178-200
            for p in (cand, os.path.join(workdir, cand)):
                if os.path.exists(p):
                    fn = p
                    break
            if fn:
                break
        if fn is None:
            raise RuntimeError("Cannot find volume file.")
        big = ds.get("endian", "LITTLE_ENDIAN") == "BIG_ENDIAN"
        dt = dtype.newbyteorder(">" if big else "<")
        volume = np.fromfile(fn, dtype=dt, count=nx * ny * nz, offset=int(ds.get("offset", 0))).astype(dtype, copy=False).reshape(nz, ny, nx)
    spacing = (1.0, 1.0, 1.0)
    if "scales" in ds:
        spacing = tuple(float(ds["scales"][k]) for k in "xyz")
    table = rasterize_transfer_function(jv["transferFunction"])
    color = np.concatenate([table[:, :3], np.ones((table.shape[0], 1), _F)], axis=1)        # :215-219
    opacity = table[:, 3].copy()
    if opacity[0] < _F(0.01):
        opacity[0] = 0                                                                    # :222-223
    if opacity[-1] < _F(0.01):
        opacity[-1] = 0
    if "scalarMappingRangeUnnormalized" in jv:
        r = jv["scalarMappingRangeUnnormalized"]
        vr = (_F(r["minimum"]), _F(r["maximum"]))
    elif "scalarMappingRange" in jv:
        r = jv["scalarMappingRange"]
        lo, hi = _F(r["minimum"]), _F(r["maximum"])
        if dtype in _INT_MAX:                                                             # :236-266: range x type maximum
            m = _INT_MAX[dtype]
            vr = (_F(m * float(lo)) if dtype.itemsize >= 4 else _F(int(m)) * lo, _F(m * float(hi)) if dtype.itemsize >= 4 else _F(int(m)) * hi)
        else:
            vr = (lo, hi)
    else:
        raise RuntimeError("unknown data range")
    cam = view["camera"]
    xyz = lambda d: tuple(float(d[k]) for k in "xyz")
    lights = []
    if "lightSource" in view:
        ls = view["lightSource"]
        lights.append((xyz(ls["position"]), tuple(float(ls["diffuse"][k]) for k in "rgb")))
    for ls in view.get("additionalLightSources", []):
        lights.append((xyz(ls["position"]), tuple(float(ls["diffuse"][k]) for k in "rgb")))
    if not lights:
        lights.append(((1.0, 1.0, 1.0), (1.0, 1.0, 1.0)))
    rate = _F(1.0) / _F(float(jv["sampleDistance"]))                                       # :399
    return dict(volume=volume, dtype=dtype, dims=(nx, ny, nz), grid_origin=(0.0, 0.0, 0.0), grid_spacing=spacing,
                tfn_color=color, tfn_opacity=opacity, value_range=(float(vr[0]), float(vr[1])),
                camera=(xyz(cam["eye"]), xyz(cam["center"]), xyz(cam["up"]), float(cam["fovy"])), lights=lights,
                volume_sampling_rate=float(rate))


def scene_from_file(path):
    """-> (Scene, Camera) ready for DeviceHIP.init, the Python counterpart of ovr::scene::create_json_scene"""
    from .renderer import Camera, Scene, TransferFunction
    d = read_scene(path)
    eye, at, up, fovy = d["camera"]
    cam = Camera(eye, at, up, fovy)
    scene = Scene(volume=d["volume"], grid_origin=d["grid_origin"], grid_spacing=d["grid_spacing"],
                  transfer_function=TransferFunction(color=d["tfn_color"], opacity=d["tfn_opacity"], value_range=d["value_range"]),
                  camera=cam, volume_sampling_rate=d["volume_sampling_rate"])
    return scene, cam
