"""Deterministic synthetic inputs for tests and bench.py (SURVEY.md 8d "Synthetic inputs"): the reference ships no
volume data, so volumes, transfer functions and cameras are generated from closed forms that numpy (CPU) and torch
(GPU, for the 1024^3 bench volume) evaluate identically."""
import math

import numpy as np


def _hash01_np(x, y, z):
    """top 24 bits of a pcg-style 32-bit hash of the voxel index, seed 0x9E3779B9, as float in [0,1)"""
    m = np.uint64(0xFFFFFFFF)
    s = (x.astype(np.uint64) + y.astype(np.uint64) * np.uint64(0x9E3779B1) + z.astype(np.uint64) * np.uint64(0x85EBCA77)
         + np.uint64(0x9E3779B9)) & m
    s = (s * np.uint64(747796405) + np.uint64(2891336453)) & m
    w = (((s >> ((s >> np.uint64(28)) + np.uint64(4))) ^ s) * np.uint64(277803737)) & m
    w = ((w >> np.uint64(22)) ^ w) & m
    return (w >> np.uint64(8)).astype(np.float32) / np.float32(16777216.0)


def _field(qx, qy, qz, h, xp):
    c1 = (0.5, 0.5, 0.5)
    c2 = (0.3, 0.6, 0.4)
    d1 = (qx - c1[0]) ** 2 + (qy - c1[1]) ** 2 + (qz - c1[2]) ** 2
    d2 = (qx - c2[0]) ** 2 + (qy - c2[1]) ** 2 + (qz - c2[2]) ** 2
    v = (0.55 * xp.exp(-d1 / 0.08) + 0.45 * xp.exp(-d2 / 0.02)
         + 0.25 * (0.5 + 0.5 * xp.sin(14.0 * qx) * xp.sin(11.0 * qy) * xp.sin(9.0 * qz)) + 0.05 * h)
    return xp.clip(v, 0.0, 1.0) if xp is np else v.clamp(0.0, 1.0)


def make_volume(n, dtype=np.float32, dims=None):
    """Synthetic scalar field V(x,y,z) on an n^3 (or dims = (nx, ny, nz)) grid, numpy, shape (nz, ny, nx)."""
    nx, ny, nz = dims if dims is not None else (n, n, n)
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    qx = x.astype(np.float32) / np.float32(max(nx - 1, 1))
    qy = y.astype(np.float32) / np.float32(max(ny - 1, 1))
    qz = z.astype(np.float32) / np.float32(max(nz - 1, 1))
    v = _field(qx, qy, qz, _hash01_np(x, y, z), np).astype(np.float32)
    return quantize(v, dtype)


def quantize(v, dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return v
    if dtype == np.float64:
        return v.astype(np.float64)
    if dtype == np.uint8:
        return np.rint(v * 255.0).astype(np.uint8)
    if dtype == np.uint16:
        return np.rint(v * 65535.0).astype(np.uint16)
    if dtype == np.int8:
        return (np.rint(v * 255.0) - 128).astype(np.int8)
    if dtype == np.int16:
        return (np.rint(v * 65535.0) - 32768).astype(np.int16)
    if dtype == np.uint32:
        return np.rint(v.astype(np.float64) * 4294967295.0).astype(np.uint32)
    if dtype == np.int32:
        return (np.rint(v.astype(np.float64) * 4294967295.0) - 2147483648).astype(np.int32)
    raise ValueError(dtype)


def make_volume_torch(n, device, dtype="float32", slab=32):
    """Same field, generated directly in HBM with torch (used for the 512^3 ... 2048^3 bench volumes)."""
    import torch
    tdt = {"float32": torch.float32, "uint8": torch.uint8, "uint16": getattr(torch, "uint16", torch.int16)}[dtype]
    out = torch.empty((n, n, n), dtype=tdt, device=device)
    ar = torch.arange(n, device=device, dtype=torch.int64)
    q = ar.to(torch.float32) / float(max(n - 1, 1))
    M = 0xFFFFFFFF
    for z0 in range(0, n, slab):
        z1 = min(n, z0 + slab)
        z, y, x = torch.meshgrid(ar[z0:z1], ar, ar, indexing="ij")
        s = (x + y * 0x9E3779B1 + z * 0x85EBCA77 + 0x9E3779B9) & M
        s = (s * 747796405 + 2891336453) & M
        w = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & M
        w = ((w >> 22) ^ w) & M
        h = (w >> 8).to(torch.float32) / 16777216.0
        v = _field(q[None, None, :], q[None, :, None], q[z0:z1, None, None], h, torch)
        if dtype == "float32":
            out[z0:z1] = v
        elif dtype == "uint8":
            out[z0:z1] = torch.round(v * 255.0).to(torch.uint8)
        else:
            vi = torch.round(v * 65535.0).to(torch.int32)
            out[z0:z1] = vi.to(tdt) if tdt != torch.int16 else (vi - 65536 * (vi >= 32768)).to(torch.int16)
        del z, y, x, s, w, h, v
    return out


_RAINBOW = [  # control points of tfn::TransferFunctionCore::fromRainbowMap (reference extern/tfn/core.h:636-650)
    (0.0 / 6.0, 0.0, 0.364706, 1.0), (1.0 / 6.0, 0.0, 1.0, 0.976471), (2.0 / 6.0, 0.0, 1.0, 0.105882),
    (3.0 / 6.0, 0.968627, 1.0, 0.0), (4.0 / 6.0, 1.0, 0.490196, 0.0), (5.0 / 6.0, 1.0, 0.0, 0.0), (6.0 / 6.0, 0.662745, 0.0, 1.0),
]


def rainbow_colors(n=1024):
    """piecewise-linear rainbow colour table sampled at (i+.5)/n (how updateColorMap rasterises, core.h:598-634)"""
    pos = np.array([c[0] for c in _RAINBOW], dtype=np.float64)
    rgb = np.array([c[1:] for c in _RAINBOW], dtype=np.float64)
    v = (np.arange(n) + 0.5) / n
    out = np.stack([np.interp(v, pos, rgb[:, k]) for k in range(3)], axis=1)
    return out.astype(np.float32)


def make_tfn(kind="sparse", n=1024, dtype=np.float32):
    """Returns (colors_rgb flat 3n, alphas flat (pos, alpha) 2n, value_range) in the app-side format of
    MainRenderer::set_transfer_function.  "sparse": alpha 0 below 40 %, ramp to 0.6 at 80 %; "dense": 0.9 * i/(n-1)."""
    colors = rainbow_colors(n)
    i = np.arange(n, dtype=np.float64)
    if kind == "sparse":
        lo, hi = int(round(0.4 * n)), int(round(0.8 * n))
        alpha = np.where(i < lo, 0.0, np.where(i < hi, 0.6 * (i - lo) / max(hi - lo, 1), 0.6))
    elif kind == "dense":
        alpha = 0.9 * i / max(n - 1, 1)
    elif kind == "opaque":   # every sample ends its ray (alpha 1): a frame of fixed per-ray / per-workgroup costs only
        alpha = np.ones(n)
    elif kind == "bumps":   # a few narrow iso-surface-like peaks: exercises the alpha>0 / alpha==0 divergence
        alpha = np.zeros(n)
        for c, w, a in ((0.35, 0.03, 0.3), (0.6, 0.04, 0.5), (0.85, 0.05, 0.8)):
            alpha = np.maximum(alpha, a * np.clip(1.0 - np.abs(i / (n - 1) - c) / w, 0.0, 1.0))
    else:
        raise ValueError(kind)
    pos = (i / max(n - 1, 1)).astype(np.float32)
    alphas = np.stack([pos, alpha.astype(np.float32)], axis=1).ravel()
    dtype = np.dtype(dtype)
    if dtype == np.uint8:
        vr = (0.0, 255.0)
    elif dtype == np.uint16:
        vr = (0.0, 65535.0)
    elif dtype == np.int8:
        vr = (-128.0, 127.0)
    elif dtype == np.int16:
        vr = (-32768.0, 32767.0)
    else:
        vr = (0.0, 1.0)
    return colors.ravel().astype(np.float32), alphas.astype(np.float32), vr


def make_camera(kind, n, convention="cell"):
    """front / oblique cameras of SURVEY.md 8d, centred on the volume's world-space centre."""
    c = (n - 1) / 2.0 if convention == "vertex" else n / 2.0
    C = np.array([c, c, c], dtype=np.float64)
    if kind == "front":
        eye = C + np.array([0.0, 0.0, 2.65 * n])
    elif kind == "oblique":
        d = np.array([-0.82, 0.41, 0.40])
        eye = C + 1.9 * n * d / np.linalg.norm(d)
    elif kind == "inside":
        eye = C + np.array([0.11 * n, -0.07 * n, 0.23 * n])
    elif kind.startswith("tilt"):   # the front camera swung off the z axis by <n> degrees (towards +x, a little +y)
        a = math.radians(float(kind[4:]))
        d = np.array([math.sin(a) * math.cos(math.radians(35.0)), math.sin(a) * math.sin(math.radians(35.0)), math.cos(a)])
        eye = C + 2.65 * n * d
    elif kind in _EXTRA_CAMERAS:   # the other axes and diagonals (layout / view-dependence measurements)
        d, dist, up = _EXTRA_CAMERAS[kind]
        d = np.array(d, dtype=np.float64)
        eye = C + dist * n * d / np.linalg.norm(d)
        return tuple(float(v) for v in eye), tuple(float(v) for v in C), up
    else:
        raise ValueError(kind)
    return tuple(float(v) for v in eye), tuple(float(v) for v in C), (0.0, 1.0, 0.0)


_EXTRA_CAMERAS = {
    "side": ((1.0, 0.0, 0.0), 2.65, (0.0, 1.0, 0.0)),        # rays along -x
    "top": ((0.0, 1.0, 0.0), 2.65, (0.0, 0.0, -1.0)),        # rays along -y
    "oblique_y": ((0.41, 0.82, 0.40), 1.9, (0.0, 0.0, 1.0)),  # y-dominant diagonal
    "oblique_z": ((0.40, 0.41, 0.82), 1.9, (0.0, 1.0, 0.0)),  # z-dominant diagonal
    "diagonal": ((1.0, 1.0, 1.0), 1.9, (0.0, 1.0, 0.0)),      # no dominant axis
}
CAMERAS = ("front", "oblique", "inside") + tuple(_EXTRA_CAMERAS)


def make_noise_tile(xy=64, seed=7, slices=64):
    """Synthetic blue-noise-like tile in the layout of the reference's noise files ([y][x][t], float32 in [0,1);
    ovr/common/random/blue_noise.h:44-47,95-99; the reference embeds data/noise/*.bin, which does not travel).  Every slice
    is white noise with its low spatial frequencies removed (toroidal FFT high-pass), rank-equalised to a uniform
    distribution - neighbouring pixels get dissimilar values, the property the mask and the pixel jitter want."""
    rng = np.random.default_rng(seed)
    fy = np.fft.fftfreq(xy)[:, None]
    fx = np.fft.fftfreq(xy)[None, :]
    hp = 1.0 - np.exp(-(fx * fx + fy * fy) / (2.0 * 0.18 ** 2))
    out = np.empty((xy, xy, slices), dtype=np.float32)
    for t in range(slices):
        w = rng.standard_normal((xy, xy))
        v = np.fft.ifft2(np.fft.fft2(w) * hp).real
        rank = np.empty(xy * xy, dtype=np.int64)
        rank[np.argsort(v.ravel(), kind="stable")] = np.arange(xy * xy)
        out[:, :, t] = ((rank + 0.5) / (xy * xy)).reshape(xy, xy).astype(np.float32)
    return out
