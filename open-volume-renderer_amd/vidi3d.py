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
