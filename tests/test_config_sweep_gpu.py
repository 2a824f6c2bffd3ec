"""Seeded sweep over combinations the targeted tests do not pair up: dtype x grid shape x shading x samples per pixel x
accumulation x pipeline x empty-space skipping x sparse sampling x image shard x camera x sampling rate x (round 2) volume layout x
pixel jitter x default transfer-function range - every case against
the CPU oracle (frame within the parity bar, primary sample counts exact)."""
import numpy as np
import pytest

from helpers import make_case, oracle_scene, hip_setup, hip_frame, compare

pytestmark = pytest.mark.gpu

DTYPES = [np.float32, np.uint8, np.uint16, np.int16, np.int8]


def _cases(n_cases=72, seed=20261003):
    import os
    if os.environ.get("OVR_SWEEP_SEED_OLD"):   # a one-off hunt with another seed / more cases of the round-1/2 generator
        seed, n_cases = int(os.environ["OVR_SWEEP_SEED_OLD"]), int(os.environ.get("OVR_SWEEP_CASES_OLD", n_cases))
    rng = np.random.default_rng(seed)
    rng2 = np.random.default_rng(seed + 2)   # round-2 dimensions, drawn from a second stream so that round 1's 72 cases stay as they were
    out = []
    for i in range(n_cases):
        dims = tuple(int(rng.integers(9, 44)) for _ in range(3))
        c = dict(
            dtype=DTYPES[int(rng.integers(len(DTYPES)))], dims=dims, shading=int(rng.integers(0, 3)), spp=int(rng.choice([1, 1, 2, 3])),
            frames=int(rng.choice([1, 2, 3])), pipeline=int(rng.choice([0, 1, 2])), skip=bool(rng.integers(2)), sparse=bool(rng.integers(4) == 0),
            shard=None if rng.integers(3) else (int(rng.integers(0, 3)), 3, int(rng.choice([8, 16, 24])), int(rng.choice([8, 16]))),
            cam=str(rng.choice(["front", "oblique", "inside"])), rate=float(rng.choice([0.5, 1.0, 2.0])),
            tf=str(rng.choice(["sparse", "dense", "bumps"])), size=(int(rng.integers(17, 90)), int(rng.integers(9, 70))),
            spacing=tuple(float(rng.choice([1.0, 0.5, 2.0])) for _ in range(3)), convention=int(rng.integers(2)),
        )
        # volume layout read (forced thin replicas exist for f32 / u16 only; elsewhere the renderer falls back to general), blue-noise
        # pixel jitter, and the invalid transfer-function range that falls back to the data range
        c.update(layout=int(rng2.choice([-1, 0, 1, 2])), jitter=bool(rng2.integers(4) == 0), default_range=bool(rng2.integers(5) == 0))
        out.append(pytest.param(c, id=f"{i:02d}-" + "-".join(str(c[k].__name__ if k == "dtype" else c[k]) for k in ("dtype", "shading", "spp", "pipeline", "skip", "sparse", "cam", "layout", "jitter"))))
    return out


def _cases_round3(n_cases=None, seed=20261004):
    """round 3: the quad replica (layout 3, every voxel type), the scene files' sampling rate 4 (shadow stride 0.625 voxel) and runs of frames long
    enough for the renderer to time its alternatives (ovr_hip_stats.tuning) under accumulation; OVR_SWEEP_CASES / OVR_SWEEP_SEED widen or move the sweep for a one-off hunt"""
    import os
    n_cases = n_cases or int(os.environ.get("OVR_SWEEP_CASES", "40"))
    seed = int(os.environ.get("OVR_SWEEP_SEED", seed))
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n_cases):
        rate = float(rng.choice([1.0, 2.0, 4.0]))
        small = rate == 4.0
        dims = tuple(int(rng.integers(9, 28 if small else 44)) for _ in range(3))
        c = dict(
            dtype=DTYPES[int(rng.integers(len(DTYPES)))], dims=dims, shading=int(rng.choice([0, 1, 2, 2])), spp=int(rng.choice([1, 1, 1, 2])),
            frames=int(rng.choice([1, 3, 6, 11])), pipeline=int(rng.choice([0, 0, 1, 2])), skip=bool(rng.integers(3) == 0), sparse=bool(rng.integers(5) == 0),
            shard=None if rng.integers(4) else (int(rng.integers(0, 2)), 2, 16, 16),
            cam=str(rng.choice(["front", "oblique", "inside"])), rate=rate, tf=str(rng.choice(["sparse", "dense", "dense", "bumps"])),
            size=(int(rng.integers(17, 50 if small else 90)), int(rng.integers(9, 40 if small else 70))),
            spacing=tuple(float(rng.choice([1.0, 0.5, 2.0])) for _ in range(3)), convention=int(rng.integers(2)),
            layout=int(rng.choice([-1, -1, 0, 3, 3, 1])), jitter=bool(rng.integers(6) == 0), default_range=False,
        )
        out.append(pytest.param(c, id=f"r3-{i:02d}-" + "-".join(str(c[k].__name__ if k == "dtype" else c[k]) for k in ("dtype", "shading", "rate", "frames", "pipeline", "skip", "sparse", "cam", "layout"))))
    return out


@pytest.mark.parametrize("c", _cases() + _cases_round3())
def test_config(ovr, oracle, hip_renderer_factory, c):
    case = make_case(ovr, oracle, n=max(c["dims"]), dtype=c["dtype"], tf=c["tf"], cam=c["cam"], size=c["size"], shading=c["shading"], rate=c["rate"],
                     spp=c["spp"], convention=c["convention"], dims=c["dims"], spacing=c["spacing"], tf_n=128)
    kw = {}
    noise = focus = None
    if c["default_range"]:
        case["vr"] = (1.0, -1.0)
    if c["jitter"]:
        noise = np.random.default_rng(5).random((16, 16, 64), dtype=np.float32)
        kw.update(jitter=1, noise=noise)
    if c["sparse"]:
        noise = np.random.default_rng(5).random((16, 16, 64), dtype=np.float32)
        focus = ((0.5, 0.45), 0.35, 0.15)
        kw.update(sparse=True, focus=focus, noise=noise)
    if c["shard"]:
        kw.update(shard=c["shard"])
    ref, _, cnt = oracle_scene(oracle, case, **kw).render(frames=c["frames"], accumulate=True)
    ren = hip_renderer_factory()
    ren.set_volume_layouts(2)
    ren.set_layout_choice(c["layout"])
    if c["jitter"]:
        ren.set_noise_tile(noise)
        ren.set_pixel_jitter(1)
    hip_setup(ovr, ren, case, accumulate=True, pipeline=c["pipeline"])
    ren.set_empty_space_skipping(c["skip"])
    if c["sparse"]:
        ren.set_noise_tile(noise)
        ren.set_focus(*focus)
        ren.set_sparse_sampling(True)
    if c["shard"]:
        ren.set_image_shard(*c["shard"])
    ren.commit()
    for _ in range(c["frames"]):
        ren.render()
    got = hip_frame(ovr, ren)[0]
    st = ren.stats()
    if c["shard"]:
        # pixels of foreign tiles are never written by this rank
        rank, world, tw, th = c["shard"]
        mask = np.zeros(got.shape[:2], bool)
        for tx, ty in ovr.tiles.owned_tiles(c["size"][0], c["size"][1], tw, th, rank, world):
            mask[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = True
        got = np.where(mask[..., None], got, 0.0).astype(np.float32)
        ref = np.where(mask[..., None], ref, 0.0).astype(np.float32)
    compare(oracle, got, ref, name=str(c))
    assert st.samples + st.skipped_samples == cnt.samples, c
    assert st.frame_index == c["frames"]
    assert st.tuning in (0, 1, 2)
    ren.close()
