"""Round 5: what the oracle's restatement of __powf does to every tolerated HIP <-> oracle difference.  For both modes of the
oracle (libm's powf, rounds 1-4; exp2f(y * log2f(x)), CUDA's definition of the intrinsic and the kernel's structure, round 5) on the GPU box:
  scenes   the 21 shipped scenes' transfer functions / cameras / rates (tests/test_shipped_scenes_gpu.py): shaded-sample count differences,
           colour differences on all and on visible pixels
  c1       C1's full frame (tests/test_full_size_gpu.py::test_c1_full_frame_vs_oracle): the shadow-sample difference
  sweep    the round-3 configuration sweep under the hunt seeds (OVR_DIAG_SEEDS=301,302,303 x OVR_DIAG_CASES=600): primary counts, parity bar
usage: python tests/powf_diag.py scenes c1 sweep  > gpurun_out/powf_diag.txt"""
import os
import sys
import time

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + "/tests", _R + "/oracle"]
import numpy as np  # noqa: E402
import ovr_amd as ovr  # noqa: E402
import oracle as O  # noqa: E402
from helpers import make_case, oracle_scene, hip_setup, hip_frame  # noqa: E402

MODES = (("libm", O.POWF_LIBM), ("exp2", O.POWF_EXP2_LOG2), ("det", O.POWF_DET))
q = lambda x: (np.clip(np.asarray(x, dtype=np.float32), np.float32(0.0), np.float32(1.0)) * np.float32(255.0)).astype(np.uint8).astype(np.int32)


def oracle_both(case, **kw):
    out = {}
    frames = kw.pop("frames", 1)
    for name, mode in MODES:
        old = O.set_powf_mode(mode)
        out[name] = oracle_scene(O, case, **kw).render(frames=frames, accumulate=True)
        O.set_powf_mode(old)
    return out


def scenes():
    import test_shipped_scenes_gpu as T
    tot = {m: dict(dshaded=0, shaded=0, worst_rel=0.0, worst_abs=0, dcol_all=0.0, dcol_vis=0.0, dalpha=0.0, dprim=0) for m, _ in MODES}
    for name in T.SCENES:
        case = T.scene_case(ovr, name)
        refs = oracle_both(case)
        ren = ovr.create_renderer("hip")
        hip_setup(ovr, ren, case, accumulate=True)
        ren.commit()
        ren.render()
        got, st = hip_frame(ovr, ren)[0], ren.stats()
        ren.close()
        line = f"scene {name:28s} hip shaded {st.shaded_samples:9d}"
        for m, _ in MODES:
            ref, _, cnt = refs[m]
            d = abs(int(st.shaded_samples) - int(cnt.shaded_samples))
            vis = q(ref[..., 3]) >= 1
            dall = float(np.abs(got[..., :3] - ref[..., :3]).max())
            dvis = float(np.abs(got[..., :3] - ref[..., :3])[vis].max()) if vis.any() else 0.0
            da = float(np.abs(got[..., 3] - ref[..., 3]).max())
            t = tot[m]
            t["dshaded"] += d; t["shaded"] += int(cnt.shaded_samples); t["worst_abs"] = max(t["worst_abs"], d)
            t["worst_rel"] = max(t["worst_rel"], d / max(int(cnt.shaded_samples), 1)); t["dcol_all"] = max(t["dcol_all"], dall)
            t["dcol_vis"] = max(t["dcol_vis"], dvis); t["dalpha"] = max(t["dalpha"], da); t["dprim"] += abs(int(st.samples) - int(cnt.samples))
            line += f" | {m}: dshaded {d:6d} of {int(cnt.borderline_samples)} borderline ({d / max(int(cnt.shaded_samples), 1):.2e}) dcol all {dall:.3g} vis {dvis:.3g} dalpha {da:.3g} d8vis {int(np.abs(q(got[..., :3]) - q(ref[..., :3]))[vis].max()) if vis.any() else 0}"
        print(line, flush=True)
    for m, _ in MODES:
        print(f"scenes total {m}: {tot[m]}", flush=True)


def c1():
    import torch
    n, size = 256, (512, 512)
    vol = ovr.synth.make_volume_torch(n, torch.device("cuda", 0), "uint8")
    colors, alphas, vr = ovr.synth.make_tfn("sparse", 1024, np.uint8)
    cam = ovr.synth.make_camera("oblique", n)
    ren = ovr.create_renderer("hip")
    ren.set_fbsize(size); ren.set_frame_accumulation(True); ren.set_shading(2); ren.set_transfer_function(colors, alphas, vr)
    ren.init(ovr.Scene(volume=vol, transfer_function=None), ovr.Camera(*cam)); ren.set_camera(*cam); ren.commit(); ren.render()
    got, st = hip_frame(ovr, ren)[0], ren.stats()
    ren.close()
    for m, mode in MODES:
        old = O.set_powf_mode(mode)
        ref, _, cnt = O.OracleScene(vol.cpu().numpy(), colors, alphas, vr, cam, size[0], size[1], shading=O.SHADE_FULL).render()
        O.set_powf_mode(old)
        print(f"c1 {m}: samples hip {st.samples} oracle {cnt.samples}; shaded {st.shaded_samples} / {cnt.shaded_samples}; shadow {st.shadow_samples} / {cnt.shadow_samples_visible} "
              f"(diff {int(st.shadow_samples) - int(cnt.shadow_samples_visible)}); max float diff {np.abs(got - ref).max():.3g}; max 8-bit {np.abs(q(got) - q(ref)).max()}", flush=True)


def sweep():
    import test_config_sweep_gpu as T
    seeds = [int(s) for s in os.environ.get("OVR_DIAG_SEEDS", "303").split(",")]
    n_cases = int(os.environ.get("OVR_DIAG_CASES", "600"))
    budget = float(os.environ.get("OVR_DIAG_BUDGET_S", "900"))
    t_start = time.time()
    for seed in seeds:
        cases = T._cases_round3(n_cases, seed)
        bad = {m: dict(prim=[], bar=[]) for m, _ in MODES}
        done = 0
        for idx, p in enumerate(cases):
            if time.time() - t_start > budget:
                break
            c = p.values[0]
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
            refs = oracle_both(case, frames=c["frames"], **kw)
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
            for _ in range(c["frames"]):
                ren.render()
            got, st = hip_frame(ovr, ren)[0], ren.stats()
            ren.close()
            for m, _ in MODES:
                ref, _, cnt = refs[m]
                g, r = got, ref
                if c["shard"]:
                    rank, world, tw, th = c["shard"]
                    mask = np.zeros(got.shape[:2], bool)
                    for tx, ty in ovr.tiles.owned_tiles(c["size"][0], c["size"][1], tw, th, rank, world):
                        mask[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = True
                    g = np.where(mask[..., None], got, 0.0).astype(np.float32); r = np.where(mask[..., None], ref, 0.0).astype(np.float32)
                if int(st.samples) + int(st.skipped_samples) != int(cnt.samples):
                    bad[m]["prim"].append((idx, int(st.samples) + int(st.skipped_samples), int(cnt.samples)))
                if np.abs(g - r).max() > 2e-4 or np.abs(q(g) - q(r)).max() > 1:
                    bad[m]["bar"].append((idx, float(np.abs(g - r).max())))
            done += 1
            if done % 50 == 0:
                print(f"sweep seed {seed}: {done} cases, {time.time() - t_start:.0f} s; mismatches so far " + "; ".join(f"{m}: primary {len(bad[m]['prim'])} bar {len(bad[m]['bar'])}" for m, _ in MODES), flush=True)
        for m, _ in MODES:
            print(f"sweep seed {seed} {m}: {done} cases; primary-count mismatches {bad[m]['prim']}; parity-bar misses {bad[m]['bar']}", flush=True)


if __name__ == "__main__":
    for part in sys.argv[1:] or ["scenes", "c1", "sweep"]:
        {"scenes": scenes, "c1": c1, "sweep": sweep}[part]()
