"""Run with OVR_HIP_LIBRARY=<repo>/open-volume-renderer_amd/libovr_hip_parity.so (the kernels built with -DOVR_PARITY_EXACT=1) - tests/test_parity_exact_gpu.py does.
Both sides then evaluate the opacity correction's __powf (shaders_raymarching.cu:64-66,118-122) with the SAME machine-independent log2 / exp2 pair
(the oracle's mode 2), and every count the parity tests tolerate a difference in must be equal EXACTLY:
  kat     the device's det pow == the oracle's, bit for bit, on 300 000 inputs
  scenes  the 21 shipped scenes (sampling rates 4 and 20: every opacity goes through the pow): primary, shaded AND shadow counts exact
  c1      C1's full frame: primary, shaded, shadow counts exact
  sweep   the configuration sweep's hunt seed 303 (600 cases, incl. #425 where the product's v_exp / v_log end a ray one step early): primary counts exact
Prints one line per part; exit code 1 on any difference."""
import os
import sys

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + "/tests", _R + "/oracle"]
import numpy as np  # noqa: E402
import ovr_amd as ovr  # noqa: E402
import oracle as O  # noqa: E402
from helpers import make_case, oracle_scene, hip_setup, hip_frame  # noqa: E402

assert ovr._lib.load().ovr_hip_built_for_exact_parity() == 1, "this check needs libovr_hip_parity.so (OVR_HIP_LIBRARY)"
O.set_powf_mode(O.POWF_DET)
q = lambda x: (np.clip(np.asarray(x, dtype=np.float32), np.float32(0.0), np.float32(1.0)) * np.float32(255.0)).astype(np.uint8).astype(np.int32)
bad = 0


def kat():
    global bad
    rng = np.random.default_rng(7)
    a = np.concatenate([rng.random(100000), 10.0 ** rng.uniform(-8, -1, 100000), 1.0 - 10.0 ** rng.uniform(-7.3, -1, 100000)]).astype(np.float32)
    x = (np.float32(1.0) - a).astype(np.float32)
    y = np.concatenate([rng.uniform(0.01, 12.0, 150000), np.full(75000, 0.25), np.full(75000, 10.0)]).astype(np.float32)
    lib = O.load()
    want = np.array([lib.ovr_oracle_det_powf(float(p), float(e)) for p, e in zip(x, y)], dtype=np.float32)
    ren = ovr.create_renderer("hip")
    got0, got1 = ren.pow_floats(x, y, 0), ren.pow_floats(x, y, 1)
    ren.close()
    n0, n1 = int((got0.view(np.uint32) != want.view(np.uint32)).sum()), int((got1.view(np.uint32) != want.view(np.uint32)).sum())
    print(f"kat: {len(x)} inputs, device (as built) != oracle: {n0}, device (det pair) != oracle: {n1}", flush=True)
    bad += n0 + n1


def scenes():
    global bad
    import test_shipped_scenes_gpu as T
    worst, ndiff, ngrad = 0.0, [], []
    for name in T.SCENES:
        case = T.scene_case(ovr, name)
        ref, rgrad, cnt = oracle_scene(O, case).render(frames=1, accumulate=True)
        ren = ovr.create_renderer("hip")
        hip_setup(ovr, ren, case, accumulate=True)
        ren.commit()
        ren.render()
        (got, grad), st = hip_frame(ovr, ren), ren.stats()
        ren.close()
        ngrad.append(int((grad.view(np.uint32) != rgrad.view(np.uint32)).sum()))
        same = (int(st.samples), int(st.shaded_samples), int(st.shadow_samples)) == (int(cnt.samples), int(cnt.shaded_samples), int(cnt.shadow_samples_visible))
        # the un-premultiplied colour of EVERY pixel, visible or not (the product's bar exempts pixels whose 8-bit alpha is 0: their colour is a ratio of
        # two sums of 6e-8 steps - equal here, because the steps are)
        d = float(np.abs(got - ref).max())
        worst = max(worst, d)
        ndiff.append(int((got.view(np.uint32) != ref.view(np.uint32)).sum()))
        if not same or d > 2e-4 or np.abs(q(got) - q(ref)).max() > 1:
            bad += 1
            print(f"scene {name}: counts hip {(st.samples, st.shaded_samples, st.shadow_samples)} oracle {(cnt.samples, cnt.shaded_samples, cnt.shadow_samples_visible)} max diff {d}", flush=True)
    print(f"scenes: {len(T.SCENES)} scenes, worst float difference on ANY pixel and channel (un-premultiplied colour included) {worst:.3g}; values that differ in any bit, per scene: RGBA {ndiff} gradient layer {ngrad}", flush=True)
    bad += sum(ndiff) + sum(ngrad)


def c1():
    global bad
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
    ref, _, cnt = O.OracleScene(vol.cpu().numpy(), colors, alphas, vr, cam, size[0], size[1], shading=O.SHADE_FULL).render()
    same = (int(st.samples), int(st.shaded_samples), int(st.shadow_samples)) == (int(cnt.samples), int(cnt.shaded_samples), int(cnt.shadow_samples_visible))
    print(f"c1: counts hip {(st.samples, st.shaded_samples, st.shadow_samples)} oracle {(cnt.samples, cnt.shaded_samples, cnt.shadow_samples_visible)} exact {same}; max float diff {np.abs(got - ref).max():.3g}, values that differ in any bit {int((got.view(np.uint32) != ref.view(np.uint32)).sum())} of {got.size}", flush=True)
    bad += 0 if same and np.abs(got - ref).max() <= 2e-4 else 1


def sweep():
    global bad
    import test_config_sweep_gpu as T
    seed, n_cases = int(os.environ.get("OVR_DETPOW_SEED", "303")), int(os.environ.get("OVR_DETPOW_CASES", "600"))
    wrong, notsame = [], []
    for idx, p in enumerate(T._cases_round3(n_cases, seed)):
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
        ref, rgrad, cnt = oracle_scene(O, case, **kw).render(frames=c["frames"], accumulate=True)
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
        got, grad = hip_frame(ovr, ren)
        st = ren.stats()
        ren.close()
        if int(st.samples) + int(st.skipped_samples) != int(cnt.samples):
            wrong.append((idx, int(st.samples) + int(st.skipped_samples), int(cnt.samples)))
        if c["shard"]:   # pixels of foreign tiles are never written by this rank
            rank, world, tw, th = c["shard"]
            mask = np.zeros(got.shape[:2], bool)
            for tx, ty in ovr.tiles.owned_tiles(c["size"][0], c["size"][1], tw, th, rank, world):
                mask[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = True
            got = np.where(mask[..., None], got, 0.0).astype(np.float32); ref = np.where(mask[..., None], ref, 0.0).astype(np.float32)
            grad = np.where(mask[..., None], grad, 0.0).astype(np.float32); rgrad = np.where(mask[..., None], rgrad, 0.0).astype(np.float32)
        nb = int((got.view(np.uint32) != ref.view(np.uint32)).sum()) + int((grad.view(np.uint32) != rgrad.view(np.uint32)).sum())
        if nb:
            notsame.append((idx, nb, float(np.abs(got - ref).max()), float(np.abs(grad - rgrad).max())))
    print(f"sweep: seed {seed}, {n_cases} cases, primary-count mismatches: {wrong}; cases whose RGBA or gradient frame differs in any bit: {len(notsame)} {notsame[:12]}", flush=True)
    bad += len(wrong) + len(notsame)


if __name__ == "__main__":
    for part in sys.argv[1:] or ["kat", "scenes", "c1", "sweep"]:
        {"kat": kat, "scenes": scenes, "c1": c1, "sweep": sweep}[part]()
    print("parity_exact_check:", "all exact" if bad == 0 else f"{bad} difference(s)", flush=True)
    sys.exit(1 if bad else 0)
