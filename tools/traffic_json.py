#!/usr/bin/env python3
"""Builds profiles/<round>_traffic.json from the summaries tools/prof.sh writes: python tools/traffic_json.py out.json KEY=profdir ...
KEY = bench.traffic_key(...) = "<config>|<camera>|<tf>|<shading>|<world>|<rate>|<fovy>|<sparse>" (what bench.py looks up).  The file is keyed
by the hash of the kernel sources it was measured for (bench.kernels_hash): bench.py quotes it only while that hash matches, else it
prints traffic: null.  Per kernel (the template instantiation with the most dispatches): mean per dispatch of
  FETCH_SIZE / WRITE_SIZE (KiB; HBM + Infinity-Cache side of L2; gfx950 tallies 128-byte read requests at 64 bytes: bytes = FETCH x 2 + WRITE,
  MI355X_MICROARCH.md), TCP_TCC_READ_REQ (L1 line fills), SQ_INSTS_VMEM_RD (gather instructions, per wave), SQ_INSTS_VALU (per wave),
  TA_TA_BUSY (summed over the 256 texture addressers), GRBM_GUI_ACTIVE (summed over the 8 XCDs) and the kernel's mean duration."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

KERNELS = ("raymarch_kernel", "shade_pool_kernel", "composite_kernel")
COUNTERS = ("FETCH_SIZE", "WRITE_SIZE", "TCP_TCC_READ_REQ_sum", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU", "TA_TA_BUSY_sum", "TD_TD_BUSY_sum", "GRBM_GUI_ACTIVE", "SQ_INSTS_LDS", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCC_HIT_sum", "TCC_MISS_sum")


def summary(path):
    per = {}   # (kernel base, instantiation) -> {counter: (dispatches, mean)}
    for line in open(os.path.join(path, "pmc_summary.txt")):
        m = re.match(r"(.*?)\s(\w+) dispatches=(\d+) mean=([0-9.e+-]+)\s*$", line)
        if not m:
            continue
        inst, ctr, n, mean = m.group(1), m.group(2), int(m.group(3)), float(m.group(4))
        base = next((k for k in KERNELS if k in inst), None)
        if base and ctr in COUNTERS:
            per.setdefault((base, inst), {})[ctr] = (n, mean)
    out = {}
    for base in KERNELS:
        cands = [(max(n for n, _ in c.values()), inst, c) for (b, inst), c in per.items() if b == base]
        if not cands:
            continue
        _, inst, c = max(cands, key=lambda t: t[0])
        out[base] = {"instantiation": inst.strip(), **{k: v[1] for k, v in c.items()}}
    stats = os.path.join(path, "kernel_stats.csv")
    if os.path.exists(stats):
        for r in csv.DictReader(open(stats)):
            for base, d in out.items():
                if d["instantiation"].split("void ")[-1] in r["Name"]:
                    d["mean_ms_rocprof"] = float(r["AverageNs"]) * 1e-6
                    d["calls"] = int(r["Calls"])
    return out


def main():
    dst, entries = sys.argv[1], {}
    for arg in sys.argv[2:]:
        key, path = arg.split("=", 1)
        if not os.path.exists(os.path.join(path, "pmc_summary.txt")):
            print("skipped (no summary):", path)
            continue
        ks = summary(path)
        fetch = {k: round(v.get("FETCH_SIZE", 0.0)) for k, v in ks.items()}
        write = {k: round(v.get("WRITE_SIZE", 0.0)) for k, v in ks.items()}
        total = sum((2 * fetch.get(k, 0) + write.get(k, 0)) * 1024 for k in KERNELS)
        entries[key] = {"fetch_size_kib": fetch, "write_size_kib": write, "traffic_bytes_per_launch": total, "kernels": ks, "profile": os.path.basename(path.rstrip("/"))}
    doc = {"_comment": "per frame per kernel, rocprofv3 --pmc in separate passes (tools/prof.sh), mean per dispatch; HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE (KiB); "
                       "valid for the kernel sources with this hash only (bench.py checks)",
           "kernels_hash": bench.kernels_hash(), "entries": entries}
    with open(dst, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", dst, "hash", doc["kernels_hash"], "entries", list(entries))


if __name__ == "__main__":
    main()
