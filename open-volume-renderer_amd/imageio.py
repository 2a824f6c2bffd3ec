"""Frame output next to the path: the file formats of the reference's `ovr::save_image`
(ovr/common/imageio.cpp:264-284: the extension picks EXR, PNG or JPG).

The pixel conversions run on the GPU (`ovr_hip_mapframe_rgba8` = image_to_rgba8, `ovr_hip_mapframe_rgba16f` = the float ->
half step of the EXR writer); this module only wraps the converted pixels into files:

* `.exr`  scan-line OpenEXR, channels A, B, G, R of type HALF, ZIP compression in blocks of 16 lines - the attributes and
          pixel type the reference asks tinyexr for (imageio.cpp:15-83).  Written here from the OpenEXR file layout; the
          reference's own `load_exr` reads these files back (tests/test_oracle_vs_ref.py).
* `.png`  8-bit RGBA (Pillow)
* `.jpg`  8-bit, quality 100 like imageio.cpp:279 (Pillow's encoder, not stb's: the files differ, the decoded pixels agree
          to JPEG accuracy; alpha is dropped as JPEG has none)
"""
import struct
import zlib

import numpy as np


def _attr(name, typ, payload):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


def _zip_block(raw: bytes) -> bytes:
    """OpenEXR ZIP: bytes de-interleaved into (even, odd) halves, delta-coded with bias 128, deflated; kept raw if not smaller"""
    a = np.frombuffer(raw, dtype=np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int16)
    d = t.copy()
    d[1:] = t[1:] - t[:-1] + 128
    comp = zlib.compress((d & 0xFF).astype(np.uint8).tobytes(), 6)
    return comp if len(comp) < len(raw) else raw


def save_exr(path, half_rgba, compression="zip", reference_channel_naming=True):
    """half_rgba: (H, W, 4) uint16 array of IEEE half bit patterns, row 0 = TOP line of the image (already flipped).

    reference_channel_naming: the reference's save_exr (imageio.cpp:27-62) reverses its channel planes for a B, G, R naming
    and then names four of them B, G, R, A - so with RGBA input the channel called "R" holds green, "G" blue, "B" alpha and
    "A" red (pinned: tests/golden/ref_probe.json, written and read back by the reference itself).  True (default) names the
    channels the same way, so a file written here loads exactly like the reference's file of the same frame; False names
    them for what they hold."""
    px = np.ascontiguousarray(half_rgba, dtype=np.uint16)
    h, w = px.shape[:2]
    lines_per_block = 16 if compression == "zip" else 1
    chlist = b""
    for name in ("A", "B", "G", "R"):  # alphabetical, as the format requires
        chlist += name.encode() + b"\0" + struct.pack("<iB3xii", 1, 0, 1, 1)  # HALF, pLinear 0, sampling 1 x 1
    chlist += b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    header = (_attr("channels", "chlist", chlist)
              + _attr("compression", "compression", bytes([3 if compression == "zip" else 0]))
              + _attr("dataWindow", "box2i", box)
              + _attr("displayWindow", "box2i", box)
              + _attr("lineOrder", "lineOrder", b"\0")  # increasing y
              + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
              + _attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
              + _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
              + b"\0")
    head = struct.pack("<II", 20000630, 2) + header
    # per scan line: the four channel planes one after the other (A, B, G, R), each w halfs
    order = [0, 3, 2, 1] if reference_channel_naming else [3, 2, 1, 0]   # which of R, G, B, A the channels A, B, G, R carry
    planes = px[:, :, order].transpose(0, 2, 1)  # (H, 4, W)
    blocks = []
    for y0 in range(0, h, lines_per_block):
        raw = planes[y0:y0 + lines_per_block].astype("<u2").tobytes()
        data = _zip_block(raw) if compression == "zip" else raw
        blocks.append(struct.pack("<ii", y0, len(data)) + data)
    table_at = len(head)
    offsets, pos = [], table_at + 8 * len(blocks)
    for b in blocks:
        offsets.append(pos)
        pos += len(b)
    with open(path, "wb") as f:
        f.write(head)
        f.write(struct.pack("<%dQ" % len(offsets), *offsets))
        for b in blocks:
            f.write(b)


def save_png(path, rgba8):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(rgba8), "RGBA").save(path)


def save_jpg(path, rgba8):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(rgba8)[:, :, :3], "RGB").save(path, quality=100, subsampling=0)
