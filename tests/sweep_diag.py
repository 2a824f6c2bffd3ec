"""Diagnoses cases of tests/test_config_sweep_gpu.py's round-3 sweep that miss the parity bar: where is the worst pixel, what are its alpha and its
premultiplied colour on both sides?   usage: OVR_SWEEP_SEED=.. OVR_SWEEP_CASES=.. python tests/sweep_diag.py <case index> ..."""
import os, sys
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + '/tests', _R + '/oracle']
import numpy as np
import ovr_amd as ovr
import oracle as O
from helpers import make_case, oracle_scene, hip_setup, hip_frame
import test_config_sweep_gpu as T

cases = T._cases_round3()
for idx in map(int, sys.argv[1:]):
    c = cases[idx].values[0]
    case = make_case(ovr, O, n=max(c["dims"]), dtype=c["dtype"], tf=c["tf"], cam=c["cam"], size=c["size"], shading=c["shading"], rate=c["rate"],
                     spp=c["spp"], convention=c["convention"], dims=c["dims"], spacing=c["spacing"], tf_n=128)
    kw = {}
    noise = focus = None
    if c["jitter"]:
        noise = np.random.default_rng(5).random((16, 16, 64), dtype=np.float32); kw.update(jitter=1, noise=noise)
    if c["sparse"]:
        noise = np.random.default_rng(5).random((16, 16, 64), dtype=np.float32); focus = ((0.5, 0.45), 0.35, 0.15); kw.update(sparse=True, focus=focus, noise=noise)
    if c["shard"]:
        kw.update(shard=c["shard"])
    for frames in sorted({1, c["frames"]}):
        ref, _, cnt = oracle_scene(O, case, **kw).render(frames=frames, accumulate=True)
        for tune in ("1", "0"):
            os.environ["OVR_HIP_TUNE"] = tune
            ren = ovr.create_renderer("hip")
            ren.set_volume_layouts(2); ren.set_layout_choice(c["layout"])
            if c["jitter"]:
                ren.set_noise_tile(noise); ren.set_pixel_jitter(1)
            hip_setup(ovr, ren, case, accumulate=True, pipeline=c["pipeline"])
            ren.set_empty_space_skipping(c["skip"])
            if c["sparse"]:
                ren.set_noise_tile(noise); ren.set_focus(*focus); ren.set_sparse_sampling(True)
            if c["shard"]:
                ren.set_image_shard(*c["shard"])
            ren.commit()
            seq = []
            for _ in range(frames):
                ren.render(); st = ren.stats(); seq.append((st.layout, st.pipeline, st.tuning))
            got = hip_frame(ovr, ren)[0]
            fbd = ovr.FrameBufferData(); ren.mapframe(fbd, device=True)
            dev_frame = fbd.rgba.data().cpu().numpy().reshape(got.shape)
            st = ren.stats()
            extra = f"host mirror == device frame: {np.array_equal(dev_frame, got)} (device frame nonzero pixels {(dev_frame[..., 3] > 0).sum()}, mirror {(got[..., 3] > 0).sum()}, oracle {(ref[..., 3] > 0).sum()}); samples hip {st.samples}+{st.skipped_samples} oracle {cnt.samples}; eye {case['cam'][0]}"
            ren.close()
            if c["shard"]:
                rank, world, tw, th = c["shard"]
                mask = np.zeros(got.shape[:2], bool)
                for tx, ty in ovr.tiles.owned_tiles(c["size"][0], c["size"][1], tw, th, rank, world):
                    mask[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = True
                got = np.where(mask[..., None], got, 0.0).astype(np.float32); r = np.where(mask[..., None], ref, 0.0).astype(np.float32)
            else:
                r = ref
            d = np.abs(got - r)
            y, x, ch = np.unravel_index(np.argmax(d), d.shape)
            pm_g, pm_r = got[..., :3] * got[..., 3:4], r[..., :3] * r[..., 3:4]
            print(f"case {idx} frames {frames} tune {tune}: max diff {d.max():.4g} at pixel ({x},{y}) ch {ch}; hip {got[y, x]} oracle {r[y, x]}; "
                  f"max alpha diff {d[..., 3].max():.3g}; max premultiplied diff {np.abs(pm_g - pm_r).max():.3g}; pixels with diff > 2e-4: {(d.max(axis=-1) > 2e-4).sum()}; "
                  f"(layout, pipeline, tuning) per frame {seq[-1]}; {extra}", flush=True)
