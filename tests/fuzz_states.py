"""Randomised walk through the renderer's state machine - the setters an interactive app calls between frames, in any order, on ONE renderer - with
every mapped frame compared against the oracle rendering the current configuration from scratch for as many frames as the device says it has
accumulated (ovr_hip_stats.frame_index).  Hunts what the scripted test_state_changes_on_one_renderer cannot: stale caches (schedule, replicas,
macrocells, request pool, measured layout / pipeline decisions, the mapped rectangle), missed or spurious accumulation resets.
usage: python tests/fuzz_states.py [episodes] [seed] [ops per episode]"""
import sys
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0] = [_R, _R + '/tests', _R + '/oracle']
import numpy as np
import ovr_amd as ovr
import oracle as O
from helpers import make_case, oracle_scene, hip_setup, hip_frame, compare

episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_ops = int(sys.argv[3]) if len(sys.argv) > 3 else 12
DTYPES = [np.float32, np.uint8, np.uint16, np.int16, np.int8]
failures = 0
LONG = __import__('os').environ.get('OVR_FUZZ_LONG') == '1'
# OVR_FUZZ_GATHER=1: every frame also goes through the N > 1 frame path with ONE rank - tiles.TileGather over gloo: pack behind the frame, the
# collective on its own stream, the scatter one call later - and the gathered frame must equal the device frame
GATHER = __import__('os').environ.get('OVR_FUZZ_GATHER') == '1'
NONFINITE = __import__('os').environ.get('OVR_FUZZ_NONFINITE') == '1'   # float volumes with a few NaN / Inf voxels
# (round 4) OVR_FUZZ_GROUP=N: the renderer is an in-process device group of N members on device 0 (ovr_hip_create_group: tiles dealt over the
# members, gathered on the leader) - every frame must still be the oracle's WHOLE frame; the shard transition becomes a tile-size change.
# OVR_FUZZ_LAZY=1: the default layouts mode - replicas are built in the background when a frame first asks for one - instead of all at upload
GROUP = int(__import__('os').environ.get('OVR_FUZZ_GROUP', '0'))
LAZY = __import__('os').environ.get('OVR_FUZZ_LAZY') == '1'
if GATHER:
    import torch
    import torch.distributed as dist
    __import__('os').environ.setdefault('MASTER_ADDR', '127.0.0.1'); __import__('os').environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('gloo', rank=0, world_size=1)


def episode(ep):
    rng = np.random.default_rng(seed * 100003 + ep)
    dims = tuple(int(rng.integers(9, 40)) for _ in range(3))
    spacing = tuple(float(rng.choice([1.0, 0.5, 2.0])) for _ in range(3))
    st8 = dict(dtype=DTYPES[int(rng.integers(len(DTYPES)))], dims=dims, spacing=spacing, tf=str(rng.choice(["sparse", "dense", "bumps"])), cam=str(rng.choice(["front", "oblique", "inside"])),
               size=(int(rng.integers(17, 80)), int(rng.integers(9, 60))), shading=int(rng.integers(0, 3)), rate=1.0, convention=int(rng.integers(2)), tf_n=int(rng.choice([64, 128])))
    origin = tuple(float(x) for x in rng.choice([0.0, -7.5, 13.0], 3))
    case = make_case(ovr, O, n=max(dims), dtype=st8["dtype"], tf=st8["tf"], cam=st8["cam"], size=st8["size"], shading=st8["shading"], rate=st8["rate"],
                     convention=st8["convention"], dims=dims, spacing=spacing, tf_n=st8["tf_n"], origin=origin, fovy=float(rng.choice([60.0, 60.0, 35.0, 95.0])))
    st8["origin"], st8["fovy"] = origin, case["fovy"]
    noise = np.random.default_rng(5).random((16, 16, 64), dtype=np.float32)
    if NONFINITE and st8["dtype"] == np.float32:   # a few NaN / +-Inf voxels: fmaxf / fminf / clamp semantics of the reference's device build
        v = case["vol"]
        idx = rng.integers(0, v.size, 6)
        v.reshape(-1)[idx] = np.array([np.nan, np.inf, -np.inf, np.nan, 3.0e38, -3.0e38], np.float32)
    ren = ovr.create_renderer("hip", devices=[0] * GROUP) if GROUP > 1 else ovr.create_renderer("hip")
    if not LAZY:
        ren.set_volume_layouts(2)
    hip_setup(ovr, ren, case, accumulate=True, pipeline=0)
    ren.set_noise_tile(noise)
    gat = [None]

    def new_gatherer():
        if GATHER:
            ren.set_image_shard(0, 1, 16, 16); ren.commit()   # the tile size the payload slots are cut for (bench.py does the same)
            ren.sync()
            w, h = case["size"]
            gat[0] = ovr.tiles.TileGather(ren, w, h, 16, 0, 1, torch.device("cuda", 0))

    def render_one():
        if not GATHER:
            ren.render()
            return
        ren.render_async(); gat[0].run(); ren.sync()
        try:
            gat[0].check()
        except RuntimeError:          # the frame overflowed its pool after its tiles were packed and was rendered again: gather the good frame
            gat[0].run(); ren.sync(); gat[0].check()

    new_gatherer()
    sparse, focus, shard, skip, jitter, accumulate = False, ((0.5, 0.45), 0.35, 0.15), None, False, False, True
    log = [f"init {st8}"]

    def check(tag):
        kw = {}
        if sparse:
            kw.update(sparse=True, focus=focus, noise=noise)
        if shard:
            kw.update(shard=shard)
        if jitter:
            kw.update(jitter=1, noise=noise)
        got = hip_frame(ovr, ren)[0]
        if GATHER:
            gat[0].flush(); ren.sync(); torch.cuda.synchronize()
            assert np.array_equal(gat[0].frame.cpu().numpy().view(np.uint32), got.view(np.uint32)), tag + " gathered frame != rendered frame"
        fbd = ovr.FrameBufferData()
        ren.mapframe(fbd, device=True)   # the host mirror is a copy of the device frame - of ALL of it, whatever rectangle was refreshed
        for lay, host in (("rgba", got), ("grad", hip_frame(ovr, ren)[1])):
            dev_frame = getattr(fbd, lay).data().cpu().numpy().reshape(host.shape)
            assert np.array_equal(dev_frame.view(np.uint32), host.view(np.uint32)), tag + f" host mirror != device frame ({lay})"
        st = ren.stats()
        ref, _, cnt = oracle_scene(O, case, **kw).render(frames=int(st.frame_index), accumulate=accumulate)
        if rng.integers(4) == 0:   # the 8-bit frame the device converts (image_to_rgba8), flipped like the PNG writer wants it
            raw = hip_frame(ovr, ren)[0]
            want = O.rgba8(raw, flip=True)
            assert np.array_equal(np.array(ren.mapframe_rgba8(flip_vertical=True), copy=True).reshape(want.shape), want), tag + " rgba8"
        if shard:
            rank, world, tw, th = shard
            mask = np.zeros(got.shape[:2], bool)
            for tx, ty in ovr.tiles.owned_tiles(case["size"][0], case["size"][1], tw, th, rank, world):
                mask[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = True
            got = np.where(mask[..., None], got, 0.0).astype(np.float32)
            ref = np.where(mask[..., None], ref, 0.0).astype(np.float32)
        try:
            compare(O, got, ref, name=tag)
        except AssertionError as e:
            # the known ill-conditioning of the reference's un-premultiplied output: a pixel with alpha of a few thousandths
            # may miss the float bar in `color / alpha` while alpha and the premultiplied colour - what the pixel shows - agree
            dd = np.abs(got - ref)
            pm = np.abs(got[..., :3] * got[..., 3:4] - ref[..., :3] * ref[..., 3:4])
            off = dd[..., :3].max(axis=-1) > 2e-4
            tolerated = not np.isnan(got).any() and dd[..., 3].max() <= 2e-4 and pm.max() <= 2e-4 and (ref[..., 3][off] < 0.02).all() and dd[..., :3].max() <= 2e-3
            d = np.abs(got - ref)
            d = np.where(np.isnan(d), np.inf, d)
            y, x, ch = np.unravel_index(np.argmax(d), d.shape)
            if not tolerated:
                raise AssertionError(f"{e}; worst pixel ({x},{y}) channel {ch}: hip {got[y, x]} oracle {ref[y, x]}; pixels over 2e-4: {(d.max(axis=-1) > 2e-4).sum()}; "
                                     f"layout {st.layout} pipeline {st.pipeline} shading {case['shading']}")
        assert st.samples + st.skipped_samples == cnt.samples, (tag, st.samples, st.skipped_samples, cnt.samples)

    try:
        render_one()
        check("initial")
        for k in range(n_ops):
            op = int(rng.integers(0, 19))
            if op == 0:
                kind = str(rng.choice(["front", "oblique", "inside", "random"]))
                if kind == "random":
                    c = np.array(case["origin"]) + np.array(dims) * np.array(spacing) / 2.0
                    d = rng.normal(size=3); d /= np.linalg.norm(d)
                    if rng.integers(3) == 0:
                        d = np.eye(3)[int(rng.integers(3))] * (1 if rng.integers(2) else -1)   # exactly along an axis
                    eye = tuple(c + d * float(rng.uniform(0.3, 2.5)) * np.linalg.norm(c) * 2 + rng.normal(size=3) * float(rng.choice([0.0, 3.0])))
                    up = (0.0, 1.0, 0.0) if abs(d[1]) < 0.9 else (0.0, 0.0, 1.0)
                    case["cam"] = (eye, tuple(c), up)
                else:
                    case["cam"] = make_case(ovr, O, n=max(dims), dtype=st8["dtype"], cam=kind, dims=dims, spacing=spacing, convention=case["convention"], tf_n=64, origin=case["origin"])["cam"]
                ren.set_camera(ovr.Camera(*case["cam"], case["fovy"])); log.append(f"camera {kind} {case['cam']}")
            elif op == 1:
                case["size"] = (int(rng.integers(9, 90)), int(rng.integers(5, 70))); ren.set_fbsize(case["size"]); log.append(f"fbsize {case['size']}")
            elif op == 2:
                tf, n = str(rng.choice(["sparse", "dense", "bumps"])), int(rng.choice([16, 64, 128, 1024]))
                case["colors"], case["alphas"], case["vr"] = ovr.synth.make_tfn(tf, n, st8["dtype"])
                ren.set_transfer_function(case["colors"], case["alphas"], case["vr"]); log.append(f"tf {tf} {n}")
            elif op == 3:
                case["shading"] = int(rng.integers(0, 3)); ren.set_shading(case["shading"]); log.append(f"shading {case['shading']}")
            elif op == 4:
                p = int(rng.integers(0, 3)); ren.set_shading_pipeline(p); log.append(f"pipeline {p}")
            elif op == 5:
                l = int(rng.integers(-1, 4)); ren.set_layout_choice(l); log.append(f"layout {l}")
            elif op == 6:
                case["rate"] = float(rng.choice([0.5, 1.0, 2.0, 4.0])); ren.set_volume_sampling_rate(case["rate"]); log.append(f"rate {case['rate']}")
            elif op == 7:
                skip = bool(rng.integers(2)); ren.set_empty_space_skipping(skip); log.append(f"skip {skip}")
            elif op == 8:
                sparse = bool(rng.integers(2))
                if sparse:
                    focus = ((float(rng.uniform(0.2, 0.8)), float(rng.uniform(0.2, 0.8))), float(rng.uniform(0.1, 0.5)), float(rng.uniform(0.02, 0.3)))
                    ren.set_focus(*focus)
                ren.set_sparse_sampling(sparse); log.append(f"sparse {sparse} {focus}")
            elif op == 9 and GATHER:
                log.append("render only (no shard changes under the gather)")
            elif op == 9 and GROUP > 1:
                tw, th = int(rng.choice([4, 8, 16, 24])), int(rng.choice([4, 8, 16]))
                ren.set_image_shard(0, 1, tw, th); log.append(f"group tile size {tw}x{th}")
            elif op == 9:
                shard = None if rng.integers(2) else (int(rng.integers(0, 2)), 2, int(rng.choice([8, 16])), int(rng.choice([8, 16])))
                ren.set_image_shard(*(shard or (0, 1, 16, 16))); log.append(f"shard {shard}")
                # the pixels of tiles this rank does not own keep what they had: only owned tiles are compared (check)
            elif op == 10:
                # (not under sparse sampling: a sparse frame writes only its sampled pixels, so after a swap the other set shows the pixels of the
                # frames rendered into IT - the reference's two sets behave the same way, device_impl.cpp:226-239 - while the oracle models one set)
                if not sparse:
                    ren.swap(); log.append("swap")
                else:
                    log.append("render only (no swap under sparse sampling)")
            elif op == 11:
                log.append("commit only")
            elif op == 13:
                case["spp"] = int(rng.choice([1, 1, 2, 3])); ren.set_sample_per_pixel(case["spp"]); log.append(f"spp {case['spp']}")
            elif op == 14:
                jitter = bool(rng.integers(2)); ren.set_pixel_jitter(1 if jitter else 0); log.append(f"blue-noise jitter {jitter}")
            elif op == 15:
                accumulate = bool(rng.integers(3) != 0); ren.set_frame_accumulation(accumulate); log.append(f"accumulate {accumulate}")
            elif op == 17:   # the same camera with another field of view
                case["fovy"] = float(rng.choice([20.0, 45.0, 60.0, 100.0]))
                ren.set_camera(ovr.Camera(*case["cam"], case["fovy"])); log.append(f"fovy {case['fovy']}")
            elif op == 18:   # a transfer-function range that covers part of the data, or an invalid one (-> the data range, volume.cpp:131-145)
                lo0, hi0 = ovr.synth.make_tfn("dense", 8, st8["dtype"])[2]
                a, b = sorted(rng.uniform(0.0, 1.0, 2))
                if rng.integers(4) == 0:   # an invalid range after a valid one KEEPS the range in effect (set_value_range ignores it, volume.cpp:135)
                    ren.set_transfer_function(case["colors"], case["alphas"], (1.0, -1.0)); log.append(f"value range (1, -1): stays {case['vr']}")
                else:
                    case["vr"] = (float(lo0 + a * (hi0 - lo0)), float(lo0 + max(b, a + 0.05) * (hi0 - lo0)))
                    ren.set_transfer_function(case["colors"], case["alphas"], case["vr"]); log.append(f"value range {case['vr']}")
            elif op == 16:   # a new volume of another shape and type on the same renderer (the app's "open file")
                dims2 = tuple(int(rng.integers(9, 40)) for _ in range(3))
                dt2 = DTYPES[int(rng.integers(len(DTYPES)))]
                new = make_case(ovr, O, n=max(dims2), dtype=dt2, tf=str(rng.choice(["sparse", "dense", "bumps"])), cam="oblique", size=case["size"], shading=case["shading"], rate=case["rate"],
                                spp=case["spp"], convention=case["convention"], dims=dims2, spacing=case["spacing"], tf_n=64, origin=case["origin"], fovy=case["fovy"])
                case.update(new); st8["dtype"] = dt2; dims = dims2
                ren.set_transfer_function(case["colors"], case["alphas"], case["vr"])
                ren.init(ovr.Scene(volume=case["vol"], grid_origin=case["origin"], grid_spacing=case["spacing"], transfer_function=None, volume_sampling_rate=case["rate"]),
                         ovr.Camera(*case["cam"], case["fovy"]))
                ren.set_sparse_sampling(sparse)
                log.append(f"new volume {dims2} {dt2.__name__}")
            else:
                log.append("render only")
            ren.commit()
            if op == 1:
                new_gatherer()
            frames = int(rng.choice([1, 1, 2, 3, 5] if not LONG else [1, 3, 14, 36]))   # LONG: past the 12-frame re-measurement and the 32-frame skipping probe
            for _ in range(frames):
                render_one()
            log.append(f"  rendered {frames}: frame_index {ren.stats().frame_index} layout {ren.stats().layout} pipeline {ren.stats().pipeline} tuning {ren.stats().tuning}")
            check(f"episode {ep} op {k}")
    except AssertionError as e:
        print(f"FAILED episode {ep} (seed {seed}): {str(e)[:600]}")
        for line in log:
            print("   ", line)
        return 1
    finally:
        ren.close()
    return 0


for ep in range(episodes):
    failures += episode(ep)
print(f"{episodes} episodes, {failures} failed", flush=True)
sys.exit(1 if failures else 0)
