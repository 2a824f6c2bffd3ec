#!/usr/bin/env python3
"""bench.py - headline benchmark of the OVR ray-marching path on MI355X.

Metric (BASELINE.json): Msamples/s (+ fps) on a 1024^3 float32 volume at 1920x1080, gradient shading + shadow march +
early-ray termination (the reference's live ray marcher), frame accumulation on - the configuration the reference's own
`renderbatch` loop measures (apps/main_batch.cpp:254-289: 5 warm-up + 25 timed blocking render() calls).

A "step" is one blocking render() of one frame.  `value` counts primary marching-loop iterations ("samples")
of all ranks per second of wall time, with the volume already resident in HBM.  With --gpus N > 1 the image plane is cut
into tiles dealt over the ranks (one process per GPU, volume replicated) and every step ends with the gather of the
tiles to rank 0 over RCCL - total work is fixed, so scaling is "strong".

`python bench.py --gpus N` starts its own N ranks (a child `python -m torch.distributed.run`, before this process touches
the GPU); under an external launcher (WORLD_SIZE set) it is one of the ranks.

Output (round 5): rank 0 prints ONE JSON line of < 4 KB on stdout - the contract's fields, the dominant kernel's `roofline`, `cpu_baseline`,
a four-figure summary per variant and extra leg (`compact_record`) - and writes the full record (views, variants, per-kernel counters, per-rank
tables, the device group, the extra legs) to `bench_detail.json` beside this script (`--detail-file`), to gpurun_out/ when that directory
exists, and as one `[bench detail]` line to stderr."""
import argparse
import hashlib
import itertools
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (n, dtype, width, height, shading, tf, camera, rate, spp)
    "c2": dict(n=512, dtype="float32", width=1024, height=1024, shading=0, tf="sparse", cam="oblique", rate=1.0, spp=1,
               workload="512^3 f32 volume, 1024x1024, absorption+emission (no shading), ERT"),
    "c3": dict(n=1024, dtype="float32", width=1920, height=1080, shading=2, tf="sparse", cam="oblique", rate=1.0, spp=1,
               workload="1024^3 f32 volume, 1920x1080, gradient shading + shadow march + ERT (reference ray marcher)"),
    "c3g": dict(n=1024, dtype="float32", width=1920, height=1080, shading=1, tf="sparse", cam="oblique", rate=1.0, spp=1,
                workload="1024^3 f32 volume, 1920x1080, gradient shading (no shadow march) + ERT"),
    "c4": dict(n=2048, dtype="uint16", width=1920, height=1080, shading=2, tf="sparse", cam="oblique", rate=1.0, spp=1,
               workload="2048^3 u16 volume (native u16 in HBM), 1920x1080, gradient shading + shadow march + ERT"),
    "c1": dict(n=256, dtype="uint8", width=512, height=512, shading=2, tf="sparse", cam="oblique", rate=1.0, spp=1,
               workload="256^3 u8 volume, 512x512 (the sample-scene shape of BASELINE C1; renderbatch defaults)"),
    "c5": dict(n=1024, dtype="float32", width=3840, height=2160, shading=2, tf="sparse", cam="oblique", rate=1.0, spp=1, jitter="blue",
               workload="1024^3 f32 volume, 3840x2160, progressive: every step is one frame of 1 blue-noise-jittered sample per pixel, "
                        "accumulated (64 steps = the 64-spp image; slice = frame % 64 of the noise tile)"),
    "c5tea": dict(n=1024, dtype="float32", width=3840, height=2160, shading=2, tf="sparse", cam="oblique", rate=1.0, spp=4,
                  workload="1024^3 f32 volume, 3840x2160, 4 TEA-jittered samples per pixel per frame (the reference's jitter), accumulated (16 steps = 64 spp)"),
    "tiny": dict(n=64, dtype="float32", width=256, height=256, shading=2, tf="sparse", cam="oblique", rate=1.0, spp=1,
                 workload="64^3 f32 volume, 256x256 (plumbing check, not a benchmark)"),
}
VOXEL_BYTES = {"float32": 4, "uint16": 2, "uint8": 1}
DTYPE_NAME = {"float32": "f32", "uint16": "u16", "uint8": "u8"}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
CAMERAS = ["front", "oblique", "inside", "side", "top", "oblique_y", "oblique_z", "diagonal"] + ["tilt%d" % a for a in (5, 10, 15, 20, 25, 30, 35, 45)]


def algorithmic_bytes(cfg, st, pixels, accumulate=True, grad=True):
    """bytes the algorithm asks for, before any cache.  Every primary sample reads one trilinear tap
    (8 voxels); a sample with non-zero opacity reads 3 more taps in the shaded modes; every shadow-march iteration reads
    one tap; every pixel writes 16 B RGBA (+32 B accumulation read+write, +12 B gradient layer)."""
    vb = VOXEL_BYTES[cfg["dtype"]]
    tap = 8 * vb
    b = st["samples"] * tap + st["shadow_samples"] * tap
    if cfg["shading"] != 0:
        b += st["shaded_samples"] * 3 * tap
    b += pixels * (16 + (32 if accumulate else 0) + (12 if grad else 0))
    return b


def nominal_bytes(cfg, st, pixels, accumulate=True, grad=True):
    """the same with SURVEY 8(d)'s nominal F (4 taps for EVERY primary sample in the shaded modes)"""
    vb = VOXEL_BYTES[cfg["dtype"]]
    tap = 8 * vb
    f = 1 if cfg["shading"] == 0 else 4
    return st["samples"] * f * tap + st["shadow_samples"] * tap + pixels * (16 + (32 if accumulate else 0) + (12 if grad else 0))


def layout_bytes(dtype, n, layout):
    """bytes of ONE resident layout of an n^3 volume (ovr_hip_kernels.hip::volume_layout): macro blocks of (cells_x x 32 x 32) cells,
    each storing bricks_x * (cells per brick + 1 apron) * 32 * 32 voxels; general = 3 (u8: 7) cells per brick along the pair
    axis, thin = 1"""
    if layout == 0:
        cells, stored = (28, 4 * 8) if dtype == "uint8" else (30, 10 * 4)
    elif layout == 3:
        cells, stored = 32, 32 * 4   # quad: every cell stores its 2 x 2 (x, y) voxels
    else:
        cells, stored = 32, 32 * 2
    m = -(-n // 32)
    return -(-n // cells) * m * m * stored * 32 * 32 * VOXEL_BYTES[dtype]


def kernels_hash():
    """hash of the DEVICE sources of libovr_hip.so (kernels, their launch code and parameter structures; round 3: not the host state
    machine ovr_hip_api.cpp any more - a fix to mapframe must not disown every counter profile): a committed PMC measurement is only
    quoted for the kernels it was taken from"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "open-volume-renderer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def traffic_key(config, cam, tf, shading, world, rate, fovy, sparse):
    return f"{config}|{cam}|{tf}|{shading}|{world}|{float(rate)}|{float(fovy)}|{int(bool(sparse))}"


N_CU, N_SIMD = 256, 1024
PHASE_EVERY = 4            # timed_leg: the steps whose frames carry per-phase events
GATHER_LINES_PER_S = 48.5e9   # 128-byte lines per second this chip gathers from a 16 GiB table, one random line per lane or per four lanes (tools/gather_granule.cpp,
                              # profiles/r04_notes.md 13; 57e9 from the 256 MB memory-side cache): the ceiling of a scattered read, 6.2 of the 8 TB/s
GATHER_CLK = 16.0      # texture-addresser clocks per 64-lane gather instruction whose quads each stay in one line (tools/ubench_lines.hip)
L1_LOOKUPS_PER_CLK = 1.4   # TCP_TOTAL_CACHE_ACCESSES per clock and CU: the most any kernel of this repository has been measured at (C5's shade kernel 1.36, fovy 45's 1.29,
                           # C5's march 1.24, C2 1.19: profiles/r05_traffic.json) - an empirical ceiling of the texture path, not a data-sheet figure
VALU_CLK = 4.0         # issue clocks of a VALU instruction of one wave (MI355X_MICROARCH.md; tools/ubench_valu.hip)


def kernel_bound(kms, abytes, ctr):
    """Which resource bounds a kernel, from its PMC counters (profiles/*_traffic.json, mean per dispatch) and its measured duration.
    Utilisations, each against the ceiling of its own unit: hbm = (FETCH_SIZE x 2 + WRITE_SIZE) / t / 8 TB/s (the L2's memory side, Infinity-
    Cache hits included); ta = gather instructions x 16 clk / (256 CUs x clocks of the launch): the instruction rate of the texture
    addressers; valu = vector instructions x 4 clk / (1024 SIMDs x clocks).  The bound is the busiest of the three.  `hbm` is only named
    when the memory side really moves at least the algorithmic bytes; a kernel whose algorithmic bytes exceed its traffic is served by
    the caches and its HBM fraction says nothing (no HBM fraction above 1)."""
    if not ctr or kms <= 0 or "GRBM_GUI_ACTIVE" not in ctr:
        return None
    clocks = ctr["GRBM_GUI_ACTIVE"] / 8.0                        # per XCD, over the profiled launch
    t_prof = ctr.get("mean_ms_rocprof", kms) * 1e-3
    traffic = (2.0 * ctr.get("FETCH_SIZE", 0.0) + ctr.get("WRITE_SIZE", 0.0)) * 1024.0
    u = {"hbm": traffic / t_prof / (HBM_PEAK_GBS * 1e9),
         "ta": ctr.get("SQ_INSTS_VMEM_RD", 0.0) * GATHER_CLK / (N_CU * clocks),
         "valu": min(ctr.get("SQ_INSTS_VALU", 0.0) * VALU_CLK / (N_SIMD * clocks), 1.0)}
    # (round 5) the L1's line lookups - one per quad of a gather and 128-byte line it touches, against L1_LOOKUPS_PER_CLK: what the texture path
    # really spends (the instruction rate above prices every quad at ONE line; tools/ubench_align.hip: 1 / 2 / 4 lines cost a quad 1.1 / 2.8 / 4.4 clocks)
    if ctr.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        u["l1"] = min(ctr["TCP_TOTAL_CACHE_ACCESSES_sum"] / (N_CU * clocks) / L1_LOOKUPS_PER_CLK, 1.0)
    # ranked against what each unit can really deliver: a gather kernel's memory side is up against the RANDOM-LINE rate (48.5 G lines/s = 6.2 of the 8 TB/s,
    # tools/gather_granule.cpp), so the headline's march - 0.60 of the nominal HBM peak, 0.77 of that ceiling, 0.67 of the L1's lookup rate - stays a memory-
    # bound kernel (halving its lookups, as the row loads did for C4's march, moves it by 2-4 %); achieved / peak / frac are still quoted against 8 TB/s
    rank = dict(u)
    rank["hbm"] = max(u["hbm"], 2.0 * ctr.get("FETCH_SIZE", 0.0) * 1024.0 / 128.0 / t_prof / GATHER_LINES_PER_S)
    bound = max(rank, key=lambda k: rank[k])
    # (15 % tolerance: the headline's march moves 7.1 GB for 7.6 GB of algorithmic bytes - the caches serve 7 % of them and the memory side
    # is still what it waits for; at 4K or at sampling rate 4 the algorithmic bytes are 2.5 ... 10 x the traffic)
    if bound == "hbm" and abytes > traffic * 1.15:
        bound = max((k for k in u if k != "hbm"), key=lambda k: u[k])
    out = {"bound": bound, "utilisation": {k: round(v, 4) for k, v in u.items()}, "traffic": traffic, "clock_ghz": clocks / t_prof / 1e9,
           "l1_fill_bytes": ctr.get("TCP_TCC_READ_REQ_sum", 0.0) * 128.0, "ta_busy": ctr.get("TA_TA_BUSY_sum", 0.0) / (N_CU * clocks)}
    # the memory side's read requests (one per 128-byte line: FETCH_SIZE x 2 KiB / 128) against the measured random-line ceiling: what a gather
    # kernel is really up against (C3 march 0.77, C4 march 0.84 - while their byte fractions read 0.65 and 0.23)
    lines = 2.0 * ctr.get("FETCH_SIZE", 0.0) * 1024.0 / 128.0
    out["gather_lines"] = {"read_requests_per_launch": lines, "g_lines_per_s": lines / t_prof / 1e9, "ceiling_g_lines_per_s": GATHER_LINES_PER_S / 1e9,
                           "frac": lines / t_prof / GATHER_LINES_PER_S,
                           "note": "ceiling = RANDOM lines from HBM; a kernel whose requests share DRAM pages or hit the 256 MB memory-side cache (the shade kernel: neighbouring bricks) can exceed it - 57 G/s from that cache, ~62 G/s streaming"}
    if bound == "ta":
        out.update(achieved=ctr["SQ_INSTS_VMEM_RD"] / t_prof / 1e9, peak=N_CU * (clocks / t_prof) / GATHER_CLK / 1e9, unit="G gather instr/s")
    elif bound == "valu":
        out.update(achieved=ctr["SQ_INSTS_VALU"] / t_prof / 1e9, peak=N_SIMD * (clocks / t_prof) / VALU_CLK / 1e9, unit="G vector instr/s")
    elif bound == "l1":
        out.update(achieved=ctr["TCP_TOTAL_CACHE_ACCESSES_sum"] / t_prof / 1e9, peak=N_CU * (clocks / t_prof) * L1_LOOKUPS_PER_CLK / 1e9, unit="G L1 line lookups/s")
    if bound != "hbm":
        # (the 4-clock price is v_fma_f32's; v_add / v_mul / v_mov issue in 2.4-2.8 clocks, tools/ubench_valu.hip: a kernel full of them can
        # exceed the nominal peak - the fraction is capped, the utilisation says "saturated")
        out["frac"] = min(out["achieved"] / out["peak"], 1.0)
    return out


def load_traffic(key):
    """measured HBM traffic (rocprofv3 PMC: FETCH_SIZE x 2 + WRITE_SIZE, collected by tools/prof.sh + tools/traffic_json.py)
    of this configuration - only if the committed profile was taken from the kernels that are running now"""
    src = None
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if name.endswith("_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    doc = json.load(f)
            except (OSError, ValueError):
                continue
            if doc.get("kernels_hash") != kernels_hash():
                src = src or f"null: profiles/{name} was measured for other kernels (hash {doc.get('kernels_hash')}), re-run tools/prof.sh"
                continue
            ent = doc.get("entries", {}).get(key)
            if ent:
                w = key.split("|")[4]
                what = "this configuration" if w == "1" else f"rank 0's image shard of {w} ranks, profiled on one GPU without the gather: bench.py --shard-of {w}"
                return ent["traffic_bytes_per_launch"], ent.get("kernels", {}), f"profiles/{name} (rocprofv3 PMC of {what}; kernels hash {doc['kernels_hash']})"
            src = src or f"null: profiles/{name} has no entry for {key}"
    return None, {}, src or "null: no committed PMC profile"


def cpu_baseline(cfg, vol_host, colors, alphas, vr, cam, noise=None, budget_s=10.0):
    """The oracle (kind "port": this repo's CPU restatement of the reference's ray marcher - the reference's own CPU device
    is OSPRay, which is not installed) timed on the host cores on a bounded sample of the same workload: the same scene at
    full resolution where a frame fits half the budget (the 256-thread host of the GPU box: ~4 s per frame), else at 1/2 x 1/2 or
    1/4 x 1/4 of the resolution (same camera and aspect), whole frames for >= 10 s, persistent thread
    pool, 8x8-pixel work items.  `value` is measured doing the work the GPU does (gradient taps + shadow march only for
    samples with opacity > 0 - bit-identical frames, tests/test_oracle_kat.py); `reference_work` is the reference's literal
    loop (a shadow march for EVERY sample), one frame at 1/8 x 1/8."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    cores = os.cpu_count() or 1
    jitter = 1 if cfg.get("jitter") == "blue" else 0

    def scene(div, skip):
        w, h = max(cfg["width"] // div, 8), max(cfg["height"] // div, 8)
        return O.OracleScene(vol_host, colors, alphas, vr, cam, w, h, fovy=60.0, spp=cfg["spp"], rate=cfg["rate"], shading=cfg["shading"],
                             skip_zero_opacity=skip, jitter=jitter, noise=noise), w, h

    sc, w, h = scene(4, True)
    sc.render(frames=1, accumulate=False, nthreads=cores, want_grad=True)   # starts the pool, pages the volume in
    # (round 4) the sample is the WHOLE frame where the host can render it inside the budget: one more quarter-resolution
    # frame is timed, and the largest of full / half / quarter resolution whose frame is predicted to take at most half the budget is used
    tq = time.perf_counter()
    sc.render(frames=1, accumulate=False, nthreads=cores, want_grad=True)
    tq = time.perf_counter() - tq
    div = 1 if 16.0 * tq <= 0.5 * budget_s else 2 if 4.0 * tq <= 0.5 * budget_s else 4
    if div != 4:
        sc, w, h = scene(div, True)
    t0 = time.perf_counter()
    frames, cnt = 0, None
    while True:
        _, _, cnt = sc.render(frames=1, accumulate=False, nthreads=cores, want_grad=True)
        frames += 1
        if time.perf_counter() - t0 > budget_s or frames >= 4096:
            break
    dt = time.perf_counter() - t0
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            models = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")]
        if models:
            cpu_model = f"{models[0]} ({len(models)} hardware threads)"
    except OSError:
        pass
    out = {"value": cnt.samples * frames / dt / 1e6, "unit": "Msamples/s", "cores": cores, "cpu_model": cpu_model, "kind": "port", "work": "same as GPU",
           "fps_equivalent_full_frame": frames / dt / float(div * div),
           "sample": f"{frames} frame(s) of the same scene at {w}x{h} (" + ("every ray of the frame" if div == 1 else f"1/{div * div} of the {cfg['width']}x{cfg['height']} rays")
                     + f"), {cores} host threads, {dt:.1f} s"}
    if cfg["shading"] == 2:
        sc2, w2, h2 = scene(8, False)
        t1 = time.perf_counter()
        _, _, c2 = sc2.render(frames=1, accumulate=False, nthreads=cores, want_grad=True)
        dt2 = time.perf_counter() - t1
        out["reference_work"] = {"value": c2.samples / dt2 / 1e6, "unit": "Msamples/s",
                                 "sample": f"1 frame at {w2}x{h2}, shadow march for every sample like the reference's loop "
                                           f"({c2.shadow_samples / max(c2.samples, 1):.0f} shadow iterations per sample), {dt2:.1f} s"}
    return out


COMPACT_LIMIT = 4096   # bytes of the ONE stdout line (the driver's record keeps a few KB of stdout; 20 KB of detail made it unparseable)


def _sig(x, digits=5):
    """floats of the stdout line: `digits` significant digits (the detail file keeps full precision)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def _leg_summary(leg, bounds=False):
    """one extra / variant leg in the stdout line: what it took, what it delivered, and the bound of its dominant kernel"""
    if not isinstance(leg, dict):
        return None
    if "error" in leg:
        return {"error": str(leg["error"])[:120]}
    r = leg.get("roofline") or {}
    dom = (r.get("kernel") or leg.get("kernel") or "").split(" ")[0]
    kd = (r.get("kernels") or {}).get(dom, {})
    out = {"ms_per_step": _sig(leg.get("ms_per_step")), "value": _sig(leg.get("value", (leg.get("gsamples_per_s") or 0.0) * 1e3 if leg.get("gsamples_per_s") is not None else None)),
           "bound": r.get("bound", leg.get("bound")), "frac": _sig(r.get("frac", leg.get("frac"))),
           "traffic_ratio": _sig(kd.get("traffic_ratio", leg.get("traffic_ratio")))}
    v = kd.get("gather_line_rate_frac")
    if v is not None:
        out["gather_line_rate_frac"] = _sig(v)
    if bounds:   # (C4: what bounds it beside its byte fraction - the unique bytes it has to move, VERDICT r4 #5 - and where its upload went, #6)
        out["compulsory_floor_bytes"] = r.get("compulsory_floor_bytes")
        u = r.get("upload_ms") or {}
        out["upload_ms"] = {k.replace("_ms", ""): round(float(x), 1) for k, x in u.items()} or None
    g = leg.get("device_group")
    if isinstance(g, dict):   # the in-process device group: which way the tiles travelled and what the leader's thread spent on launching + shipping a frame
        out["gather"] = g.get("gather")
        out["host_us_per_frame"] = _sig(g.get("host_us_per_frame"))
    return out


def compact_record(d, detail_name):
    """The ONE stdout line (< COMPACT_LIMIT bytes): the contract's fields, the dominant kernel's roofline, the CPU baseline and a summary per extra leg.
    Everything else - views, variants, per-kernel notes, per-rank tables, the device group - is in the detail file (and on stderr).  Protocol mirrored:
    apps/main_batch.cpp:278-289 (5 warm-up + N timed blocking render() calls, fps = N / wall)."""
    r = d.get("roofline", {})
    dom = (r.get("kernel") or "").split(" ")[0]
    kd = (r.get("kernels") or {}).get(dom, {})
    c = d.get("config", {})
    out = {k: _sig(d.get(k)) for k in ("metric", "value", "unit", "fps", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {k: c.get(k) for k in ("workload", "name", "volume", "image", "camera", "transfer_function", "sampling_rate", "spp", "shading", "volume_layout_read", "parallelism") if k in c}
    pf = d.get("per_frame", {})
    out["per_frame"] = {k: _sig(pf.get(k), 10) for k in ("samples", "shaded_samples", "shadow_samples", "active_pixels") if k in pf}
    out["roofline"] = {"kernel": r.get("kernel"), "bound": r.get("bound"), "achieved": _sig(r.get("achieved")), "peak": r.get("peak"), "unit": r.get("unit"), "frac": _sig(r.get("frac")),
                       "traffic": r.get("traffic"), "traffic_ratio": _sig(kd.get("traffic_ratio")), "kernel_ms": _sig(r.get("kernel_ms")),
                       "algorithmic_bytes_per_launch": r.get("algorithmic_bytes_per_launch"),
                       "gather_line_rate_frac": _sig((kd.get("gather_lines") or {}).get("frac")),
                       "frame": {"frac": _sig((r.get("pipeline") or {}).get("frac")), "kernel_ms": _sig((r.get("pipeline") or {}).get("kernel_ms")),
                                 "algorithmic_bytes": (r.get("pipeline") or {}).get("algorithmic_bytes_per_launch")},
                       "kernels_ms": {k: _sig(v.get("ms")) for k, v in (r.get("kernels") or {}).items()},
                       "kernels_frac": {k: _sig(v.get("frac")) for k, v in (r.get("kernels") or {}).items()},
                       "compulsory_floor_bytes": r.get("compulsory_floor_bytes"), "upload_ms": r.get("upload_ms")}
    if "cpu_baseline" in d:
        cb = d["cpu_baseline"]
        out["cpu_baseline"] = {k: _sig(cb.get(k)) for k in ("value", "unit", "cores", "cpu_model", "kind", "sample")}
    if d.get("without_phase_events"):
        out["ms_per_step_without_phase_events"] = _sig(d["without_phase_events"].get("ms_per_step"))
    if d.get("with_empty_space_skipping"):
        sk = d["with_empty_space_skipping"]
        out["with_empty_space_skipping"] = {"ms_per_step": _sig(sk.get("ms_per_step")), "frames_bit_identical": sk.get("frames_bit_identical")}
    var = r.get("variants") or {}
    if var:
        out["variants"] = {k: {kk: vv for kk, vv in _leg_summary(v).items() if kk in ("ms_per_step", "bound", "frac")} for k, v in var.items()}
    if d.get("extra"):
        out["extra"] = {k: _leg_summary(v, bounds=k.startswith("c4")) for k, v in d["extra"].items()}
    for k in ("rccl_ranks", "backend", "device_count", "gather"):
        if k in d:
            out[k] = d[k]
    if d.get("ranks"):
        rk = d["ranks"]
        out["ranks"] = {k: {m: _sig(rk[k][m], 4) for m in ("min", "mean", "max")} for k in ("kernel_ms", "gather_ms", "gather_wait_ms", "step_ms") if k in rk}
        out["ranks"]["work_imbalance_max_over_mean"] = _sig(rk.get("work_imbalance_max_over_mean"), 4)
    if d.get("device_group"):
        g = d["device_group"]
        out["device_group"] = {"devices": g.get("devices"), "gather": g.get("gather"), "host_us_per_frame": _sig(g.get("host_us_per_frame")),
                               "kernel_ms_per_member": [_sig(v, 4) for v in (g.get("per_member") or {}).get("kernel_ms", [])][:16]}
    out["detail"] = detail_name
    line = json.dumps(out, separators=(",", ":"))
    # never above the limit: shed the optional blocks, least important first (the contract's fields, roofline and cpu_baseline stay)
    for k in ("device_group", "ranks", "variants", "with_empty_space_skipping", "per_frame", "extra"):
        if len(line) < COMPACT_LIMIT:
            break
        out.pop(k, None)
        out["shed"] = out.get("shed", []) + [k]
        line = json.dumps(out, separators=(",", ":"))
    return line


def write_detail(d, path):
    """the full record: beside the script (or --detail-file), under gpurun_out/ when that exists (merged back from the GPU box), and on stderr"""
    text = json.dumps(d)
    paths = [path]
    g = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(g) and os.path.dirname(os.path.abspath(path)) == ROOT:
        paths.append(os.path.join(g, os.path.basename(path)))
    for p in paths:
        try:
            with open(p, "w") as f:
                f.write(text + "\n")
        except OSError as e:
            print(f"[bench] could not write {p}: {e}", file=sys.stderr)
    print("[bench detail] " + text, file=sys.stderr)


def run_extra_leg(argv, timeout_s):
    """one more configuration as a child `python bench.py ...` (its own process: its own volume, nothing shared with the timed region that
    has already finished), trimmed to the figures the record needs.  Never raises: a failure is reported in place of the figures."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE",
                                                               "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "OVR_BENCH_FORCE_GATHER", "OVR_BENCH_CONFIG")}
    import tempfile
    fd, detail = tempfile.mkstemp(prefix="ovr_bench_leg_", suffix=".json")
    os.close(fd)
    cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--no-extras", "--no-cpu-baseline", "--no-views", "--no-skip-leg", "--detail-file", detail]
    t0 = time.perf_counter()
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            so, se = p.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, 9)   # exactly the process group this call started
            except OSError:
                pass
            p.communicate()
            return {"command": " ".join(cmd[1:]), "error": f"timed out after {timeout_s} s"}
        if p.returncode != 0:
            return {"command": " ".join(cmd[1:]), "error": f"exit code {p.returncode}: {(se or '')[-600:]}"}
        with open(detail) as f:   # the child's full record (its stdout carries the compact line only)
            doc = json.load(f)
    except Exception as e:  # noqa: BLE001 - an extra leg must never take the headline down
        return {"command": " ".join(cmd[1:]), "error": repr(e)[:600]}
    finally:
        try:
            os.unlink(detail)
        except OSError:
            pass
    r = doc.get("roofline", {})
    out = {"command": " ".join(cmd[1:]), "wall_s": round(time.perf_counter() - t0, 1), "workload": doc["config"]["workload"], "parallelism": doc["config"].get("parallelism"),
           "value": doc["value"], "unit": doc["unit"], "fps": doc["fps"], "ms_per_step": doc["ms_per_step"], "steps": doc["steps"], "dtype": doc["dtype"],
           "samples_per_frame": doc["per_frame"]["samples"],
           "roofline": {"kernel": r.get("kernel"), "bound": r.get("bound"), "frac": r.get("frac"), "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": r.get("unit"),
                        "traffic": r.get("traffic"), "hbm_algorithmic_frac": r.get("hbm_algorithmic_frac"),
                        "kernels": {k: dict({f: v.get(f) for f in ("ms", "bound", "frac", "hbm_algorithmic_frac", "traffic_ratio", "utilisation")}, gather_line_rate_frac=(v.get("gather_lines") or {}).get("frac"))
                                    for k, v in r.get("kernels", {}).items()},
                        "volume_upload_ms": r.get("volume_upload_ms"), "upload_ms": r.get("upload_ms"), "volume_resident_bytes": r.get("volume_resident_bytes"),
                        "compulsory_floor_bytes": r.get("compulsory_floor_bytes"), "compulsory_floor_ms": r.get("compulsory_floor_ms")}}
    if "without_phase_events" in doc:
        out["ms_per_step_without_phase_events"] = doc["without_phase_events"]["ms_per_step"]
    if "device_group" in doc:
        out["device_group"] = doc["device_group"]
    return out


def launch_ranks(args):
    """parent of a multi-GPU run: start one process per GPU and wait.  Nothing here touches HIP or torch.cuda."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # (torch.distributed.run's own parser takes a bare `--n` for an abbreviation of its options, even behind the script's name)
    fwd = ["--edge" if a == "--n" else a for a in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + fwd
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)   # the reference's renderbatch times 25 frames ...
    ap.add_argument("--warmup", type=int, default=5)   # ... after 5 warm-up frames (apps/main_batch.cpp:278-289)
    ap.add_argument("--config", default=os.environ.get("OVR_BENCH_CONFIG", "c3"), choices=sorted(CONFIGS))
    ap.add_argument("--camera", default=None, choices=CAMERAS)
    ap.add_argument("--tf", default=None, choices=["sparse", "dense", "bumps", "opaque"])
    ap.add_argument("--shading", type=int, default=None, choices=[0, 1, 2])
    ap.add_argument("--dtype", default=None, choices=sorted(VOXEL_BYTES), help="voxel type override (exploration: the named configurations fix it)")
    ap.add_argument("--n", "--edge", dest="n", type=int, default=None, help="volume edge override (exploration; --edge is the same option)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tile", type=int, default=16, help="image-shard tile size in pixels (16: best balance over 8 ranks, tools/shard_balance.py)")
    ap.add_argument("--no-skip-leg", action="store_true", help="do not time the extra leg with empty-space skipping (N = 1 only)")
    ap.add_argument("--no-views", action="store_true", help="do not time the camera x transfer-function matrix (N = 1 only)")
    ap.add_argument("--layout", type=int, default=-1, choices=[-1, 0, 1, 2, 3], help="volume layout a frame reads: -1 automatic (default), 0 general, 1 thin, 2 thin transposed, 3 quad")
    ap.add_argument("--pipeline", type=int, default=0, choices=[0, 1, 2], help="shading pipeline: 0 automatic, 1 in place, 2 pooled")
    ap.add_argument("--lds-staging", action="store_true", help="unshaded march of float volumes: stage the bricks of each round through LDS (measurement switch)")
    ap.add_argument("--skip-empty", action="store_true", help="enable macrocell empty-space skipping (not the headline: fewer samples are fetched)")
    ap.add_argument("--rate", type=float, default=None, help="volume sampling rate override (the scene files say 4: serializer_vidi3d.cpp:402; renderbatch's default is 1)")
    ap.add_argument("--fovy", type=float, default=60.0, help="vertical field of view (renderbatch renders 60: renderer.h:149-152; the scene files say 45)")
    ap.add_argument("--gather-every", type=int, default=1, help="N > 1: gather the tiles to rank 0 on every K-th step only (and at the end of the timed region) - progressive "
                    "accumulation that is displayed when it is mapped (SURVEY 8e on C5); the default gathers every frame, as an interactive display would")
    ap.add_argument("--shard-of", type=int, default=0, help="ONE process renders rank --shard-rank's image shard of this many ranks, without a gather: a profilable stand-in for one rank of the N-GPU run (its counters are filed under world = N)")
    ap.add_argument("--shard-rank", type=int, default=0)
    ap.add_argument("--sparse-sampling", action="store_true", help="the foveated mode with the interactive app's default focus (apps/main_app.cpp:123-124)")
    ap.add_argument("--devices", default=None, help="ONE process drives these HIP devices as an in-process device group (ovr_hip_create_group - what the C++ plugin does "
                    "for OVR_HIP_DEVICES / --hip-devices): image tiles over the devices, gathered on the first one over RCCL or peer copies; e.g. 0,1,2,3 (0,0: a rehearsal on one card)")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "bench_detail.json"), help="where rank 0 writes the full record (views, variants, per-kernel counters, per-rank tables, "
                    "extra legs); stdout carries one compact line of < 4 KB")
    ap.add_argument("--no-extras", action="store_true", help="do not run the extra legs of the default command (C4 and C5 on one GPU, rank 0's shard of 8, the device group)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}")
    worker(args, world)


def worker(args, world):
    # stdout carries exactly ONE line, the JSON: libraries that print banners there (RCCL's version block at the first collective, gloo's
    # connection messages) are routed to stderr for the lifetime of the worker
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import ovr_amd as ovr

    cfg = dict(CONFIGS[args.config])
    if args.camera:
        cfg["cam"] = args.camera
    if args.tf:
        cfg["tf"] = args.tf
    if args.shading is not None:
        cfg["shading"] = args.shading
    if args.rate is not None:
        cfg["rate"] = args.rate
    if args.dtype or args.n:
        cfg["dtype"] = args.dtype or cfg["dtype"]
        cfg["n"] = args.n or cfg["n"]
        cfg["workload"] += f" [overridden: {cfg['n']}^3 {cfg['dtype']}]"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # OVR_BENCH_FORCE_GATHER=1 (under torch.distributed.run with ONE rank): take the N > 1 frame path - process group, image shard of
    # world 1, pack, RCCL gather on the communication stream, scatter - on a single GPU; a smoke test of the RCCL plumbing
    multi = world > 1 or (os.environ.get("OVR_BENCH_FORCE_GATHER") == "1" and "RANK" in os.environ)
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # "nccl" IS RCCL on ROCm.  OVR_BENCH_BACKEND=gloo + OVR_BENCH_ONE_GPU=1 rehearse the N > 1 path on a one-GPU box
        backend = os.environ.get("OVR_BENCH_BACKEND", "nccl")
        if os.environ.get("OVR_BENCH_ONE_GPU") == "1":
            local_rank = 0
        n_dev = torch.cuda.device_count()   # (counting devices does not initialise the GPU)
        if local_rank >= n_dev:
            # a fresh process that has not touched the GPU: say what is wrong and leave - the launcher ends the other ranks
            raise SystemExit(f"bench.py: rank {rank} has LOCAL_RANK {local_rank} but this node shows {n_dev} GPU(s) "
                             f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')}, ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES')})")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    n, W, H = cfg["n"], cfg["width"], cfg["height"]
    np_dtype = {"float32": np.float32, "uint16": np.uint16, "uint8": np.uint8}[cfg["dtype"]]
    vol = ovr.synth.make_volume_torch(n, dev, cfg["dtype"])
    tfns = {cfg["tf"]: ovr.synth.make_tfn(cfg["tf"], 1024, np_dtype)}
    colors, alphas, vr = tfns[cfg["tf"]]
    cam = ovr.synth.make_camera(cfg["cam"], n)
    noise = ovr.synth.make_noise_tile(64) if cfg.get("jitter") == "blue" else None

    group_devices = [int(d) for d in args.devices.split(",")] if args.devices else None
    if group_devices and multi:
        raise SystemExit("bench.py: --devices is the single-process device group; it does not combine with --gpus N > 1 ranks")
    ren = ovr.create_renderer("hip", local_rank, devices=group_devices)
    # the call sequence of the reference's renderbatch (apps/main_batch.cpp:254-276)
    ren.set_fbsize((W, H))
    ren.set_frame_accumulation(True)
    ren.set_sample_per_pixel(cfg["spp"])
    ren.set_volume_sampling_rate(cfg["rate"])
    ren.set_shading(cfg["shading"])
    ren.set_empty_space_skipping(args.skip_empty)
    ren.set_transfer_function(colors, alphas, vr)
    if noise is not None:
        ren.set_noise_tile(noise)
        ren.set_pixel_jitter(ovr.JITTER_BLUE_NOISE)
    if multi:
        ren.set_image_shard(rank, world, args.tile, args.tile)
    elif args.shard_of > 1:
        ren.set_image_shard(args.shard_rank, args.shard_of, args.tile, args.tile)
    key_world = args.shard_of if (args.shard_of > 1 and not multi) else world   # the world the counters of profiles/*_traffic.json are filed under
    ren.set_layout_choice(args.layout)
    ren.set_shading_pipeline(args.pipeline)
    ren.set_lds_staging(args.lds_staging)
    scene = ovr.Scene(volume=vol, transfer_function=None, volume_sampling_rate=cfg["rate"])
    torch.cuda.synchronize()
    t_up = time.perf_counter()
    ren.init(scene, ovr.Camera(*cam))   # ovr_hip_set_volume (re-layout into bricks; on every device of a group) + the first commit
    volume_upload_ms = (time.perf_counter() - t_up) * 1e3
    resident_after_upload = int(ren.volume_info().resident_bytes)
    # (round 5) what ovr_hip_set_volume itself spent where: allocation (a FRESH hipMalloc: 30-60 ms per GiB - C4's 21.5 GB in a new process),
    # copies into the device (none for this device array), kernels (re-bricking, macrocell ranges, data range)
    upload_split = {k: round(v, 3) for k, v in ren.upload_times().items()}

    def set_cam():
        if args.fovy == 60.0:
            ren.set_camera(*cam)  # fovy 60, as renderbatch ends up with (renderer.h:149-152)
        else:
            ren.set_camera(ovr.Camera(*cam, fovy=args.fovy))

    set_cam()
    ren.set_sparse_sampling(False)
    if args.sparse_sampling:
        # the foveated mode of the interactive app with its default focus (apps/main_app.cpp:123-124,184)
        if noise is None:
            ren.set_noise_tile(ovr.synth.make_noise_tile(64))
        ren.set_focus((0.5, 0.5), 0.06, 0.07)
        ren.set_sparse_sampling(True)
    ren.commit()
    vinfo = ren.volume_info()
    vol_host = None
    want_cpu = world == 1 and not multi and not args.no_cpu_baseline   # rank 0 at N = 1 only
    if want_cpu:
        vol_host = vol.cpu().numpy() if cfg["dtype"] != "uint16" or hasattr(torch, "uint16") else vol.cpu().numpy().view(np.uint16)
    del vol
    torch.cuda.empty_cache()

    gatherer = None
    if multi:
        gatherer = ovr.tiles.TileGather(ren, W, H, args.tile, rank, world, dev, time_every=4)   # the steps beside the rendering are timed on every 4th pair of frames

    def step():
        if not multi:
            ren.render()          # blocking, like the reference's render() (optix7/device.cpp:35-43)
            step.count += 1
        else:
            # one frame = march/shade/composite of this rank's tiles, then the gather of all tiles to rank 0.  The gather of
            # frame i runs on its own stream while frame i+1 renders (tiles.TileGather); the host waits once per frame
            ren.render_async()
            step.count += 1
            if step.count % max(args.gather_every, 1) == 0:
                gatherer.run()
                step.gathered = step.count
            ren.sync()
            gatherer.check()
    step.count = step.gathered = 0

    def timed_leg(steps, warmup):
        """warmup untimed steps, then exactly `steps` steps between barrier + device-synchronise pairs"""
        for _ in range(warmup):
            step()
        # a shade-heavy configuration makes the renderer time its alternatives (layout, pipeline: ovr_hip_stats.tuning == 1) for up to a
        # dozen frames: they belong to the warm-up, not to the timed steps (untimed extra steps, counted in `extra_warmup`)
        # ... and so do the frames rendered from the general layout while the replica the camera asks for is still being built in the
        # background (ovr_hip_stats.replicas_building).  N > 1: every rank probes / builds on its own - the ranks agree on one more step
        # while ANY of them is still at it (those frames used to land in the timed steps of the multi-GPU line)
        def unsettled(expired):
            # (every rank makes the same number of these calls: the loop below ends on the REDUCED flag and on counters all ranks share - a
            # rank whose own patience has run out keeps answering "settled" instead of leaving the collective)
            st = ren.stats()
            mine = (st.tuning == 1 or st.replicas_building > 0) and not expired
            if dist is None:
                return mine
            flag = torch.tensor([1 if mine else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            return bool(flag.item())
        extra, t_settle = 0, time.perf_counter()
        while extra < 64 and step.count > 0 and unsettled(time.perf_counter() - t_settle > 3.0):
            step()
            extra += 1
            if ren.stats().replicas_building > 0:
                time.sleep(0.002)   # a replica is being re-bricked on the side stream: let it have the memory system for a moment
        if extra:
            for _ in range(warmup):   # the first frames on a replica that has just become resident page it in: warm-up again
                step()
                extra += 1
        timed_leg.extra_warmup = extra
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        tot = dict(samples=0, shaded_samples=0, shadow_samples=0, rays=0, active_pixels=0, skipped_samples=0, skipped_shadow_samples=0)
        leg = dict(kernel_ms=0.0, phase_ms=[0.0, 0.0, 0.0], last=None)
        # per-phase times (two more events between the frame's kernels, ~16 us a frame) are taken on every PHASE_EVERY-th step of the region and scaled:
        # the per-kernel durations are still HIP events over the timed region, the region pays a quarter of what they cost
        phases_on = timed_leg.phases
        n_phase = 0
        t0 = time.perf_counter()
        for i in range(steps):
            sampled = phases_on and i % PHASE_EVERY == 0
            if phases_on:
                ren.set_phase_timing(sampled)
            step()
            st = ren.stats()
            for k in tot:
                tot[k] += getattr(st, k)
            leg["kernel_ms"] += st.kernel_ms
            if sampled:
                n_phase += 1
                leg["phase_ms"][0] += st.march_ms
                leg["phase_ms"][1] += st.shade_ms
                leg["phase_ms"][2] += st.composite_ms
            leg["last"] = st
        if phases_on:
            ren.set_phase_timing(True)
            leg["phase_ms"] = [p * steps / max(n_phase, 1) for p in leg["phase_ms"]]   # as sums over `steps` frames, like kernel_ms
            leg["phase_steps_sampled"] = n_phase
        if gatherer is not None:
            if step.gathered != step.count:   # --gather-every: the last frame has not been gathered yet
                gatherer.run()
                step.gathered = step.count
            gatherer.flush()   # the last frame's tiles reach rank 0's frame inside the timed region
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        leg["dt"] = time.perf_counter() - t0
        leg["tot"] = tot
        return leg

    def kernel_report(c, per_launch, ph, k_ms, pooled, traffic_by_kernel, pool_chunks=0):
        """per kernel: algorithmic bytes of the taps that kernel executes / its own mean launch duration (HIP events around
        each kernel, recorded inside libovr_hip.so on the renderer's stream; profiles/*/kernel_stats.csv agrees)"""
        tap = 8 * VOXEL_BYTES[c["dtype"]]
        pixels = per_launch["active_pixels"]
        abytes = algorithmic_bytes(c, per_launch, pixels)
        fb_bytes = pixels * (16 + 32 + 12)
        if pooled and c["spp"] > 1:
            # one march/shade/composite pass per sample-per-pixel generation: the events bracket the whole sequence
            parts = [("raymarch pipeline (%d generations of march -> shade -> composite)" % c["spp"], abytes, k_ms)]
        elif pooled:
            parts = [("raymarch_kernel", per_launch["samples"] * tap, ph[0]),
                     ("shade_pool_kernel", (3 * per_launch["shaded_samples"] + per_launch["shadow_samples"]) * tap, ph[1]),
                     ("composite_kernel", fb_bytes, ph[2])]
        else:
            parts = [("raymarch_kernel", abytes, ph[0])]
        kern = {}
        for kname, kb, kms in parts:
            if kms > 0:
                ctr = traffic_by_kernel.get(kname)
                # the in-place kernel and the pooled march are both raymarch_kernel: a profile of the other pipeline does not describe this one
                if ctr and kname == "raymarch_kernel":
                    m = re.search(r"raymarch_kernel<\d+, \d+, \d+, (true|false)", ctr.get("instantiation", ""))
                    if not m or (m.group(1) == "true") != bool(pooled):
                        ctr = None
                kb_ = kernel_bound(kms, kb, ctr)
                kern[kname] = {"ms": kms, "algorithmic_bytes_per_launch": kb, "bound": "hbm", "achieved": kb / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": kb / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "traffic_ratio": None}
                if kb_:
                    # traffic_ratio = counter bytes / algorithmic bytes: well above 1 = lines fetched for a fraction of their voxels (C4: rays 3-5 voxels
                    # apart, one 128-byte brick per 2 x 2 footprint - 2.85), below 1 = the caches serve part of the algorithmic bytes
                    kern[kname].update(traffic=kb_["traffic"], traffic_ratio=(kb_["traffic"] / kb if kb > 0 else None), utilisation=kb_["utilisation"], gather_lines=kb_["gather_lines"],
                                       l1_fill_bytes=kb_["l1_fill_bytes"], ta_busy=kb_["ta_busy"], clock_ghz=kb_["clock_ghz"], hbm_algorithmic_frac=kern[kname]["frac"])
                    if kb_["bound"] != "hbm":
                        kern[kname].update(bound=kb_["bound"], achieved=kb_["achieved"], peak=kb_["peak"], unit=kb_["unit"], frac=kb_["frac"])
                else:
                    kern[kname]["bound_note"] = "no PMC profile of this configuration for these kernels: HBM assumed"
                    if kern[kname]["frac"] > 1.0:
                        # more algorithmic bytes than HBM can deliver: the caches serve them and HBM is not the bound - which unit is, only
                        # the counters can say (tools/prof.sh); no fraction is printed rather than an HBM fraction above 1
                        kern[kname].update(bound="cache-served (no counter profile)", hbm_algorithmic_frac=kern[kname]["frac"], frac=None, achieved=None, peak=None, unit=None)
        if "composite_kernel" in kern:
            # not algorithmic bytes but what this pipeline makes the kernel read besides the framebuffer: every request slot of the
            # frame's chunks (32 B each) - the reason its measured traffic is ~3 x its framebuffer bytes
            kern["composite_kernel"]["request_bytes_read"] = int(pool_chunks) * 2048
        dom = max(kern, key=lambda k: kern[k]["ms"]) if kern else None
        return kern, dom, abytes

    timed_leg.phases = True
    main_leg = timed_leg(args.steps, args.warmup)
    main_extra_warmup = timed_leg.extra_warmup
    dt, tot, kernel_ms, phase_ms, last_stats = main_leg["dt"], main_leg["tot"], main_leg["kernel_ms"], main_leg["phase_ms"], main_leg["last"]
    # The same steps again without the two events between the frame's kernels (ovr_hip_set_phase_timing(0), ABI v9: what the plugin runs).  The
    # headline's timed region keeps them - its per-kernel durations ARE those events - and pays ~16 us per frame for it; this leg is never `value`.
    ren.set_phase_timing(False)
    timed_leg.phases = False
    plain_leg = timed_leg(args.steps, 2)
    timed_leg.phases = True
    ren.set_phase_timing(True)
    without_phase_events = {"ms_per_step": plain_leg["dt"] / args.steps * 1e3, "fps": args.steps / plain_leg["dt"],
                            "msamples_per_s": plain_leg["tot"]["samples"] / plain_leg["dt"] / 1e6, "steps": args.steps,
                            "note": "ovr_hip_set_phase_timing(0): no hipEventRecord between the frame's kernels - the plugin's setting; same frames (the headline's region takes them on every 4th step)"}
    if multi and world == 1:
        # forced single-rank gather: the gathered frame must be the renderer's own frame
        fb = ovr.FrameBufferData()
        ren.mapframe(fb, device=True)
        ok = bool(torch.equal(gatherer.frame, fb.rgba.data()))
        print(f"[bench] forced gather on one rank over '{dist.get_backend()}': gathered frame == rendered frame: {ok}", file=sys.stderr)
        if not ok:
            raise SystemExit("gathered frame differs from the rendered frame")

    # extra leg (N = 1): the same frames with empty-space skipping over the reference's macrocell grids; reported
    # beside the headline, never as `value`.  Same number of accumulated frames after a reset, so the frames must be equal bit for bit
    skip_leg = None
    if not multi and not args.skip_empty and not args.no_skip_leg:
        fb = ovr.FrameBufferData()
        set_cam()   # any camera commit resets the accumulation (device_impl.cpp:125-144)
        ren.commit()
        for _ in range(args.warmup + args.steps):
            ren.render()
        ren.mapframe(fb, device=True)
        plain = fb.rgba.data().clone()
        ren.set_empty_space_skipping(True)
        set_cam()
        ren.commit()
        for _ in range(args.warmup):
            ren.render()
        # (with most samples skipped the frame is shade-heavy: the renderer measures its alternatives, the quad replica is built for it -
        # warm-up, like in timed_leg; the frame count of the accumulation is made up to the plain leg's below)
        n_settle, t_settle = 0, time.perf_counter()
        while n_settle < 64 and time.perf_counter() - t_settle < 3.0 and (ren.stats().tuning == 1 or ren.stats().replicas_building > 0):
            ren.render()
            n_settle += 1
            if ren.stats().replicas_building > 0:
                time.sleep(0.002)
        if n_settle:
            set_cam()
            ren.commit()
            for _ in range(args.warmup):
                ren.render()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ren.render()
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        ren.mapframe(fb, device=True)
        st1 = ren.stats()
        skip_leg = {"fps": args.steps / dt1, "ms_per_step": dt1 / args.steps * 1e3, "frames_bit_identical": bool(torch.equal(plain, fb.rgba.data())),
                    "samples_fetched_per_frame": int(st1.samples), "samples_skipped_per_frame": int(st1.skipped_samples),
                    "shadow_samples_skipped_per_frame": int(st1.skipped_shadow_samples),
                    "layout": ["general", "thin", "thin transposed", "quad"][st1.layout], "pipeline": "pooled" if st1.pipeline == 2 else "in place",
                    "tuning": ["rules", "probing", "measured choice"][st1.tuning], "phase_ms_last_frame": {"march": st1.march_ms, "shade": st1.shade_ms, "composite": st1.composite_ms}}
        ren.set_empty_space_skipping(False)
        ren.commit()

    # view x transfer-function matrix (N = 1; SURVEY 8d: "both cameras and both TFs reported"): the headline cell is the timed
    # region above, the other cells are short legs of the same renderer (3 warm-up + 10 timed frames each)
    views = None
    if not multi and not args.no_views and not args.skip_empty:
        views = {}
        for vcam, vtf in itertools.product(("front", "oblique"), ("sparse", "dense")):
            if (vcam, vtf) == (cfg["cam"], cfg["tf"]):
                leg, vsteps = main_leg, args.steps
            else:
                if vtf not in tfns:
                    tfns[vtf] = ovr.synth.make_tfn(vtf, 1024, np_dtype)
                ren.set_transfer_function(*tfns[vtf])
                ren.set_camera(*ovr.synth.make_camera(vcam, n))
                ren.commit()
                vsteps = 10
                leg = timed_leg(vsteps, 3)
            pl = {k: v / vsteps for k, v in leg["tot"].items()}
            ph = [p / vsteps for p in leg["phase_ms"]]
            vc = dict(cfg, cam=vcam, tf=vtf)
            _, v_ctr, _ = load_traffic(traffic_key(args.config, vcam, vtf, cfg["shading"], key_world, cfg["rate"], args.fovy, args.sparse_sampling))
            kern, dom, abytes = kernel_report(vc, pl, ph, leg["kernel_ms"] / vsteps, leg["last"].pipeline == 2, v_ctr, leg["last"].pool_chunks)
            views[f"{vcam}/{vtf}"] = {
                "ms_per_step": leg["dt"] / vsteps * 1e3, "fps": vsteps / leg["dt"], "gsamples_per_s": leg["tot"]["samples"] / leg["dt"] / 1e9,
                "samples_per_frame": pl["samples"], "shaded_samples_per_frame": pl["shaded_samples"], "shadow_samples_per_frame": pl["shadow_samples"],
                "phase_ms": {"march": ph[0], "shade": ph[1], "composite": ph[2]},
                "kernel": dom, "bound": kern[dom]["bound"] if dom else None, "frac": kern[dom]["frac"] if dom else None, "unit": kern[dom]["unit"] if dom else None,
                "utilisation": kern[dom].get("utilisation") if dom else None, "traffic": kern[dom]["traffic"] if dom else None,
                "kernel_fracs": {k: v["frac"] for k, v in kern.items()}, "kernel_bounds": {k: v["bound"] for k, v in kern.items()},
                "layout": ["general", "thin", "thin transposed", "quad"][leg["last"].layout], "pipeline": "pooled" if leg["last"].pipeline == 2 else "in place",
                "extra_warmup": timed_leg.extra_warmup if leg is not main_leg else main_extra_warmup,
                "pipeline_frac": abytes / (leg["kernel_ms"] / vsteps * 1e-3) / 1e9 / HBM_PEAK_GBS if leg["kernel_ms"] > 0 else None}
        ren.set_transfer_function(colors, alphas, vr)
        set_cam()
        ren.commit()

    # the other settings SURVEY 8d wants beside the headline (N = 1): the frame without the shadow march, at the scene files' fovy
    # of 45 degrees (renderbatch renders 60: renderer.h:149-152), at the scene files' sampling rate 4 (sampleDistance 0.25), and the
    # sparse (foveated) sampling mode of SURVEY 8 f3
    variants = None
    if views is not None:
        variants = {}
        for vname, vkw in (("no_shadow_march", dict(shading=1)), ("fovy_45", dict(fovy=45.0)), ("sampling_rate_4", dict(rate=4.0)),
                           ("sparse_sampling", dict(sparse=True))):
            if "shading" in vkw and cfg["shading"] != 2:
                continue
            vc = dict(cfg, **{k: v for k, v in vkw.items() if k in cfg})
            ren.set_shading(vc["shading"])
            ren.set_volume_sampling_rate(vc["rate"])
            ren.set_camera(ovr.Camera(*cam, fovy=vkw.get("fovy", 60.0)))
            if vkw.get("sparse"):
                # the foveated mode of the interactive app with its default focus (apps/main_app.cpp:123-124,184): mask from the
                # noise tile (generate_mask.cu:55-120), ballot / prefix compaction, sparse launch
                if noise is None:
                    ren.set_noise_tile(ovr.synth.make_noise_tile(64))
                ren.set_focus((0.5, 0.5), 0.06, 0.07)
                ren.set_sparse_sampling(True)
            ren.commit()
            vsteps = 5 if vname == "sampling_rate_4" else 10
            leg = timed_leg(vsteps, 3)
            pl = {k: v / vsteps for k, v in leg["tot"].items()}
            ph = [p / vsteps for p in leg["phase_ms"]]
            _, v_ctr, _ = load_traffic(traffic_key(args.config, cfg["cam"], cfg["tf"], vc["shading"], key_world, vc["rate"], vkw.get("fovy", 60.0), vkw.get("sparse", False)))
            kern, dom, abytes = kernel_report(vc, pl, ph, leg["kernel_ms"] / vsteps, leg["last"].pipeline == 2, v_ctr, leg["last"].pool_chunks)
            variants[vname] = {
                "ms_per_step": leg["dt"] / vsteps * 1e3, "fps": vsteps / leg["dt"], "gsamples_per_s": leg["tot"]["samples"] / leg["dt"] / 1e9,
                "samples_per_frame": pl["samples"], "shaded_samples_per_frame": pl["shaded_samples"], "shadow_samples_per_frame": pl["shadow_samples"],
                "rendered_pixels_per_frame": pl["active_pixels"],
                "phase_ms": {"march": ph[0], "shade": ph[1], "composite": ph[2]},
                "kernel": dom, "bound": kern[dom]["bound"] if dom else None, "frac": kern[dom]["frac"] if dom else None, "unit": kern[dom]["unit"] if dom else None,
                "utilisation": kern[dom].get("utilisation") if dom else None, "traffic": kern[dom]["traffic"] if dom else None,
                "layout": ["general", "thin", "thin transposed", "quad"][leg["last"].layout], "pipeline": "pooled" if leg["last"].pipeline == 2 else "in place",
                "hbm_algorithmic_frac_frame": abytes / (leg["kernel_ms"] / vsteps * 1e-3) / 1e9 / HBM_PEAK_GBS if leg["kernel_ms"] > 0 else None}
            if vkw.get("sparse"):
                ren.set_sparse_sampling(False)
        ren.set_shading(cfg["shading"])
        ren.set_volume_sampling_rate(cfg["rate"])
        set_cam()
        ren.commit()

    # the N > 1 line explains itself: per rank the kernel phases, the steps beside the rendering, the work and the elapsed time
    rank_report = None
    if dist is not None:
        torch.cuda.synchronize()
        gt = gatherer.collect_times() if gatherer is not None else {}
        mine = torch.tensor([phase_ms[0] / args.steps, phase_ms[1] / args.steps, phase_ms[2] / args.steps, kernel_ms / args.steps,
                             gt.get("pack", 0.0), gt.get("gather", 0.0), gt.get("gather_wait", 0.0), gt.get("unpack", 0.0),
                             dt / args.steps * 1e3, tot["samples"] / args.steps, tot["shaded_samples"] / args.steps, tot["shadow_samples"] / args.steps,
                             tot["active_pixels"] / args.steps, float(torch.cuda.device_count()), float(last_stats.layout), float(last_stats.pipeline)],
                            dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allr, mine)
        if rank == 0:
            cols = ["march_ms", "shade_ms", "composite_ms", "kernel_ms", "pack_ms", "gather_ms", "gather_wait_ms", "unpack_ms", "step_ms",
                    "samples", "shaded_samples", "shadow_samples", "active_pixels", "device_count", "layout", "pipeline"]
            tab = np.array([a.cpu().numpy() for a in allr])
            rank_report = {c: {"min": float(tab[:, i].min()), "mean": float(tab[:, i].mean()), "max": float(tab[:, i].max())} for i, c in enumerate(cols[:9])}
            rank_report["per_rank"] = {c: [float(v) for v in tab[:, i]] for i, c in enumerate(cols)}
            work = tab[:, 9] + 3.0 * tab[:, 10] + tab[:, 11]   # taps: primary + 3 per shaded sample + shadow
            rank_report["work_imbalance_max_over_mean"] = float(work.max() / max(work.mean(), 1e-30))
            rank_report["kernel_imbalance_max_over_mean"] = float(tab[:, 3].max() / max(tab[:, 3].mean(), 1e-30))
            rank_report["payload_bytes_per_rank"] = int(gatherer.payload_bytes) if gatherer is not None else 0
            rank_report["gather_note"] = ("gather_ms = the collective as each rank's communication stream sees it (includes waiting for the slowest rank); "
                                          "gather_wait_ms = what the render stream stalls on it (the gather of frame i overlaps the rendering of frame i+1); "
                                          "unpack_ms is rank 0's scatter of all payloads")
    # max over ranks of the elapsed time, sum over ranks of the work
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        w = torch.tensor([tot[k] for k in sorted(tot)] + [kernel_ms], dtype=torch.float64, device=dev)
        wmax = w.clone()
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        dist.all_reduce(wmax, op=dist.ReduceOp.MAX)
        for i, k in enumerate(sorted(tot)):
            tot[k] = int(w[i].item())
        kernel_ms_max = float(wmax[-1].item())
    else:
        kernel_ms_max = kernel_ms

    if rank == 0:
        steps = args.steps
        per_step = {k: v / steps for k, v in tot.items()}
        # roofline of the dominant kernel (raymarch_kernel): algorithmic bytes per launch / mean launch duration
        # (HIP events recorded on the renderer's own stream around the launch, inside libovr_hip.so)
        k_ms = kernel_ms_max / steps
        per_launch = {k: v / world for k, v in per_step.items()}
        pixels_per_launch = per_launch["active_pixels"]
        nbytes = nominal_bytes(cfg, per_launch, pixels_per_launch)
        traffic, traffic_by_kernel, traffic_source = (None, {}, "null: the empty-space skipping leg is not profiled") if args.skip_empty else \
            load_traffic(traffic_key(args.config, cfg["cam"], cfg["tf"], cfg["shading"], key_world, cfg["rate"], args.fovy, args.sparse_sampling))
        pooled = last_stats.pipeline == 2
        ph = [p / steps for p in phase_ms]
        kern, dom, abytes = kernel_report(cfg, per_launch, ph, k_ms, pooled, traffic_by_kernel, last_stats.pool_chunks)
        achieved = abytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # compulsory floor: every brick of the layout the frame reads once + the framebuffer traffic, at the HBM peak - an
        # upper bound of the unique bytes a frame can touch; what the march alone has to read when rays are sparser than voxels
        read_bytes = layout_bytes(cfg["dtype"], n, last_stats.layout)   # the layout this frame read, not every resident replica
        floor_bytes = read_bytes + pixels_per_launch * (16 + 32 + 12)
        out = {
            "metric": "Msamples/s (primary ray-march samples after ERT); fps alongside",
            "value": tot["samples"] / dt / 1e6,
            "unit": "Msamples/s",
            "fps": steps / dt,
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "extra_warmup": main_extra_warmup,
            "ms_per_step": dt / steps * 1e3,
            "without_phase_events": without_phase_events,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": DTYPE_NAME[cfg["dtype"]],
            "data": "synthetic",
            "config": {"workload": cfg["workload"], "name": args.config, "volume": f"{n}^3 {cfg['dtype']}", "image": f"{W}x{H}",
                       "transfer_function": cfg["tf"], "camera": cfg["cam"], "fovy": args.fovy, "sparse_sampling": bool(args.sparse_sampling), "sampling_rate": cfg["rate"],
                       "spp": cfg["spp"], "pixel_jitter": "blue-noise tile (synthetic 64x64x64), slice = frame % 64" if noise is not None else "RandomTEA iff spp > 1 (reference)",
                       "shading": ["none", "gradient", "gradient+shadow"][cfg["shading"]],
                       "frame_accumulation": True, "empty_space_skipping": bool(args.skip_empty), "tuning": ["rules", "probing", "measured choice"][last_stats.tuning], "volume_layout_read": ["general", "thin", "thin transposed", "quad"][last_stats.layout], "parallelism": (f"image tiles {args.tile}x{args.tile} over {world} rank(s)" if key_world == world else f"stand-in: rank {args.shard_rank}'s shard of {key_world} ranks on one GPU, no gather")},
            "per_frame": {k: per_step[k] for k in sorted(per_step)},
            # the dominant kernel of the frame (longest mean launch); the whole pipeline and the other kernels beside it
            "roofline": {"bound": kern[dom]["bound"], "achieved": kern[dom]["achieved"], "peak": kern[dom]["peak"], "unit": kern[dom]["unit"],
                         "frac": kern[dom]["frac"], "traffic": kern[dom]["traffic"], "traffic_source": traffic_source,
                         "bound_note": "bound = the busiest of: HBM side of L2 (FETCH x 2 + WRITE vs 8 TB/s), gather-instruction rate of the texture addressers "
                                       "(16 clk per instruction and CU), the L1's line lookups (l1: one per quad and line; ceiling 1.4 per clock and CU, the most measured), vector-instruction "
                                       "issue (4 clk per instruction and SIMD), from the PMC profile of this configuration; achieved / peak / frac are in that "
                                       "resource's unit, utilisation lists all of them",
                         "utilisation": kern[dom].get("utilisation"), "hbm_algorithmic_frac": kern[dom].get("hbm_algorithmic_frac", kern[dom]["frac"]),
                         "kernel": dom + ("" if "pipeline" in dom else " (pooled pipeline: march -> shade -> composite)" if pooled else " (in-place pipeline)"),
                         "kernel_ms": kern[dom]["ms"], "algorithmic_bytes_per_launch": kern[dom]["algorithmic_bytes_per_launch"],
                         "kernels": kern,
                         "pipeline": {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": k_ms,
                                      "algorithmic_bytes_per_launch": abytes,
                                      "nominal_frac_survey_F4": (nbytes / (k_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if k_ms > 0 else 0.0},
                         "compulsory_floor_ms": floor_bytes / (HBM_PEAK_GBS * 1e9) * 1e3,
                         "compulsory_floor_bytes": floor_bytes,
                         # (round 4) ovr_hip_set_volume uploads the general layout only; replicas are built in the background when a frame asks for one
                         "volume_upload_ms": volume_upload_ms, "upload_ms": upload_split, "volume_resident_bytes_after_upload": resident_after_upload,
                         "volume_resident_bytes": int(ren.volume_info().resident_bytes), "volume_layout_read_bytes": read_bytes,
                         "phase_ms_rank0": {"march": ph[0], "shade": ph[1], "composite": ph[2]},
                         "pool_chunks": int(last_stats.pool_chunks),
                         "lds_staging": {"on": bool(args.lds_staging), "fallback_taps_per_frame": int(last_stats.lds_fallback_taps),
                                         "unstaged_workgroup_rounds_per_frame": int(last_stats.lds_unstaged_rounds), "workgroup_rounds_per_frame": int(last_stats.lds_rounds)}},
        }
        if group_devices:
            gn, gkind, gms = ren.group_info()
            members = [ren.member_stats(i) for i in range(gn)]
            out["n_gpus"] = len(set(group_devices))
            out["config"]["parallelism"] = (f"ONE process, device group {group_devices} (ovr_hip_create_group - the C++ plugin's OVR_HIP_DEVICES path): image tiles "
                                            f"16x16 over {gn} member(s), gathered on device {group_devices[0]}")
            out["device_group"] = {"devices": group_devices, "distinct_devices": len(set(group_devices)), "gather": ["none", "peer copies", "RCCL send/recv (ncclCommInitAll)"][gkind],
                                   "gather_tail_ms_last_frame": gms,
                                   "host_us_last_frame": dict(zip(("enqueue", "ship", "finish", "scatter"), (round(v, 1) for v in ren.group_host_times()))),
                                   "host_us_per_frame": round(sum(ren.group_host_times()[:2]), 1),
                                   "per_member": {"march_ms": [m.march_ms for m in members], "shade_ms": [m.shade_ms for m in members], "composite_ms": [m.composite_ms for m in members],
                                                  "kernel_ms": [m.kernel_ms for m in members], "samples": [int(m.samples) for m in members],
                                                  "layout": [m.layout for m in members], "pipeline": [m.pipeline for m in members]},
                                   "note": "value counts the samples of all members; kernel times in `roofline` are the slowest member's; a device listed twice is a rehearsal on one card"}
        if multi:
            out["rccl_ranks"] = dist.get_world_size()
            out["backend"] = dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")
            out["device_count"] = torch.cuda.device_count()
            out["gather"] = "every frame" if args.gather_every <= 1 else f"every {args.gather_every}th frame and the last one of the timed region (--gather-every)"
            out["ranks"] = rank_report
        if views is not None:
            out["roofline"]["views"] = views
        if variants:
            out["roofline"]["variants"] = variants
        if skip_leg is not None:
            out["with_empty_space_skipping"] = skip_leg
        if want_cpu:
            out["cpu_baseline"] = cpu_baseline(cfg, vol_host, colors, alphas, vr, cam, noise)
    # the timed region and everything that belongs to the headline is over: the other ranks may go, the renderer's memory is returned
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ren.close()
    if rank == 0:
        # Extra legs of the DEFAULT command, each a child process after the headline's measurements, never `value`:
        #   N = 1: the two 8-GPU configurations on this one GPU (C4, C5), rank 0's image shard of 8 (what one GPU of the 8-GPU run launches), and the
        #          in-process device group on this card (two members on device 0: the C++ plugin's multi-GPU path, exercised; its speed means nothing)
        #   N > 1: the device group over the N GPUs of this run - ONE process, the unmodified apps' path (the other ranks have left by now)
        default_cmd = (args.config == "c3" and not (args.camera or args.tf or args.dtype or args.n or args.rate or args.skip_empty or args.sparse_sampling
                                                    or args.shard_of or args.devices or args.lds_staging) and args.shading is None and args.layout == -1 and args.pipeline == 0
                       and args.fovy == 60.0 and os.environ.get("OVR_BENCH_FORCE_GATHER") != "1" and os.environ.get("OVR_BENCH_ONE_GPU") != "1")
        # rehearsal switches (one-GPU box): OVR_BENCH_FORCE_EXTRAS=1 runs the extra legs although this is not the default command (e.g. two gloo
        # ranks on one card), OVR_BENCH_EXTRA_DEVICES=0,0 replaces the N > 1 leg's device list (the box has one GPU)
        if os.environ.get("OVR_BENCH_FORCE_EXTRAS") == "1":
            default_cmd = True
        if default_cmd and not args.no_extras:
            torch.cuda.empty_cache()
            extra = {}
            if world == 1:
                extra["c4_one_gpu"] = run_extra_leg(["--config", "c4", "--steps", "10", "--warmup", "3"], 420)
                extra["c5_one_gpu"] = run_extra_leg(["--config", "c5", "--steps", "10", "--warmup", "3"], 300)
                extra["c3_shard_of_8"] = run_extra_leg(["--shard-of", "8", "--steps", "20", "--warmup", "5"], 300)
                extra["c3_device_group_rehearsal"] = run_extra_leg(["--devices", "0,0", "--steps", "10", "--warmup", "5"], 300)
            else:
                devs = os.environ.get("OVR_BENCH_EXTRA_DEVICES") or ",".join(str(i) for i in range(world))
                extra["c3_device_group"] = run_extra_leg(["--devices", devs, "--steps", str(args.steps), "--warmup", str(args.warmup)], 420)
                os.environ["OVR_HIP_GATHER"] = "copy"   # the same group with peer-to-peer copies instead of RCCL send / recv
                extra["c3_device_group_peer_copies"] = run_extra_leg(["--devices", devs, "--steps", str(args.steps), "--warmup", str(args.warmup)], 420)
                del os.environ["OVR_HIP_GATHER"]
            out["extra"] = extra
        sys.stdout.flush()
        write_detail(out, args.detail_file)
        os.write(json_fd, (compact_record(out, os.path.basename(args.detail_file)) + "\n").encode())


if __name__ == "__main__":
    main()
