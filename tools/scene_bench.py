"""Frame times on the shapes of the reference's 21 shipped scenes (tests/golden/scenes): each scene's dimensions, voxel type, spacing,
transfer function, value range and camera at 1920x1080 with renderbatch's settings (fovy 60, sampling rate 1, reference shading,
accumulation) - on a synthetic field mapped into the scene's value range, because the datasets do not ship.  With the plain kernels and
with empty-space skipping enabled (the plugin's default).   usage: python tools/scene_bench.py [scene substring]"""
import os, sys, time
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + '/tests']
import numpy as np, torch, ovr_amd as ovr

SCENES = os.path.join(_R, 'tests', 'golden', 'scenes')
pick = sys.argv[1] if len(sys.argv) > 1 else ''
dev = torch.device('cuda', 0)


def synth(dims, dtype, lo, hi):
    """the bench's field on an (nx, ny, nz) grid, mapped into [lo, hi] of the scene's type, generated slab by slab in HBM"""
    nx, ny, nz = dims
    tdt = {np.dtype('float32'): torch.float32, np.dtype('uint8'): torch.uint8, np.dtype('uint16'): getattr(torch, 'uint16', torch.int16),
           np.dtype('int16'): torch.int16, np.dtype('float64'): torch.float64}[np.dtype(dtype)]
    out = torch.empty((nz, ny, nx), dtype=tdt, device=dev)
    qx = torch.arange(nx, device=dev, dtype=torch.float32) / max(nx - 1, 1)
    qy = torch.arange(ny, device=dev, dtype=torch.float32) / max(ny - 1, 1)
    qz = torch.arange(nz, device=dev, dtype=torch.float32) / max(nz - 1, 1)
    for z0 in range(0, nz, 32):
        z1 = min(nz, z0 + 32)
        X, Y, Z = qx[None, None, :], qy[None, :, None], qz[z0:z1, None, None]
        h = torch.rand((z1 - z0, ny, nx), device=dev)
        v = ovr.synth._field(X, Y, Z, h, torch).clamp(0, 1)
        v = lo + v * (hi - lo)
        if np.dtype(dtype).kind == 'f':
            out[z0:z1] = v.to(tdt)
        else:
            info = np.iinfo(dtype)
            vi = torch.round(v.clamp(info.min, info.max)).to(torch.int32)
            out[z0:z1] = vi.to(tdt) if not (np.dtype(dtype) == np.uint16 and tdt == torch.int16) else (vi - 65536 * (vi >= 32768)).to(torch.int16)
    return out


print(f"{'scene':36s} {'dims':>16s} {'type':>7s} {'plain ms':>9s} {'skip ms':>8s} {'fps (skip)':>10s}  layout  Msamples  shaded  skipped%")
for name in sorted(f for f in os.listdir(SCENES) if f.endswith('.json') and pick in f):
    d = ovr.vidi3d.read_scene(os.path.join(SCENES, name), load_volume=False)
    dims, dtype = d['dims'], np.dtype(d['dtype'])
    lo, hi = d['value_range']
    if dtype.kind != 'f':
        info = np.iinfo(dtype)
        lo, hi = max(lo, float(info.min)), min(hi, float(info.max))
    if not hi > lo:
        lo, hi = 0.0, 1.0
    vol = synth(dims, dtype, float(lo), float(hi))
    n = len(d['tfn_opacity'])
    colors = np.ascontiguousarray(d['tfn_color'][:, :3], dtype=np.float32).ravel()
    alphas = np.stack([np.linspace(0.0, 1.0, n, dtype=np.float32), d['tfn_opacity'].astype(np.float32)], axis=1).ravel()
    eye, at, up, fovy = d['camera']
    ren = ovr.create_renderer('hip')
    ren.set_fbsize((1920, 1080)); ren.set_frame_accumulation(True); ren.set_sample_per_pixel(1); ren.set_shading(2)
    ren.set_transfer_function(colors, alphas, d['value_range'])
    ren.set_shading_pipeline(int(os.environ.get('OVR_SCENE_PIPELINE', '0')))   # 0 automatic, 1 in place, 2 pooled
    ren.set_layout_choice(int(os.environ.get('OVR_SCENE_LAYOUT', '-1')))       # -1 automatic, 0 general, 1 thin, 2 thin transposed, 3 quad
    rate = float(os.environ.get('OVR_SCENE_RATE', '1.0'))                       # renderbatch renders rate 1; the scene files say 4
    ren.set_volume_sampling_rate(rate)
    ren.init(ovr.Scene(volume=vol, grid_origin=d['grid_origin'], grid_spacing=d['grid_spacing'], transfer_function=None, volume_sampling_rate=rate), ovr.Camera(eye, at, up))
    ren.set_camera(eye, at, up)
    res = {}
    for skip in ((False, True) if os.environ.get('OVR_SCENE_SKIP_LEG', '1') != '0' else (False, False)):
        ren.set_empty_space_skipping(skip); ren.commit()
        for _ in range(int(os.environ.get('OVR_SCENE_WARMUP', '3'))):
            ren.render()
        # (round 4) replicas are built in the background and the renderer measures layout / pipeline on shade-heavy scenes: time settled frames
        t_settle, n_settle = time.perf_counter(), 0
        while n_settle < 64 and time.perf_counter() - t_settle < 5.0 and (ren.stats().tuning == 1 or ren.stats().replicas_building > 0):
            ren.render(); n_settle += 1
        if n_settle:
            for _ in range(3):
                ren.render()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            ren.render()
        torch.cuda.synchronize()
        res[skip] = ((time.perf_counter() - t0) / 10 * 1e3, ren.stats())
    res.setdefault(True, res[False])
    st0, st1 = res[False][1], res[True][1]
    tot = st1.samples + st1.skipped_samples
    print(f"{name:36s} {'x'.join(map(str, dims)):>16s} {str(dtype):>7s} {res[False][0]:9.3f} {res[True][0]:8.3f} {1e3 / res[True][0]:10.0f}  {st0.layout:6d} {st0.samples / 1e6:9.1f} {st0.shaded_samples / 1e6:7.1f} "
          f"{100.0 * st1.skipped_samples / max(tot, 1):7.1f}  kernels(skip leg) {st1.skipping_kernels}", flush=True)
    ren.close()
    del vol
    torch.cuda.empty_cache()
