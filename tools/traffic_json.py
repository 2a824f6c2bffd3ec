#!/usr/bin/env python3
"""Builds profiles/<round>_traffic.json from the PMC summaries tools/prof.sh writes: python tools/traffic_json.py out.json KEY=profdir ...
KEY = "<config>|<camera>|<tf>|<shading>|<world>" (what bench.py looks up).  The file is keyed by the hash of the kernel sources it
was measured for (bench.kernels_hash): bench.py quotes it only while that hash matches, else it prints traffic: null.
HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE, unit KiB (gfx950 counts 128-byte read requests at 64 bytes: MI355X_MICROARCH.md)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

KERNELS = ("raymarch_kernel", "shade_pool_kernel", "composite_kernel")


def summary(path):
    out = {"FETCH_SIZE": {}, "WRITE_SIZE": {}}
    for line in open(os.path.join(path, "pmc_summary.txt")):
        m = re.search(r"(raymarch_kernel|shade_pool_kernel|composite_kernel).*\s(FETCH_SIZE|WRITE_SIZE) dispatches=\d+ mean=([0-9.e+]+)", line)
        if m:
            out[m.group(2)][m.group(1)] = out[m.group(2)].get(m.group(1), 0.0) + float(m.group(3))
    return out


def main():
    dst, entries = sys.argv[1], {}
    for arg in sys.argv[2:]:
        key, path = arg.split("=", 1)
        s = summary(path)
        fetch = {k: round(v) for k, v in s["FETCH_SIZE"].items()}
        write = {k: round(v) for k, v in s["WRITE_SIZE"].items()}
        total = sum((2 * fetch.get(k, 0) + write.get(k, 0)) * 1024 for k in KERNELS)
        entries[key] = {"fetch_size_kib": fetch, "write_size_kib": write, "traffic_bytes_per_launch": total, "profile": os.path.basename(path.rstrip("/"))}
    doc = {"_comment": "HBM traffic per frame per kernel, rocprofv3 --pmc in separate passes (tools/prof.sh), mean per dispatch; bytes = FETCH_SIZE x 2 + WRITE_SIZE (KiB); "
                       "valid for the kernel sources with this hash only (bench.py checks)",
           "kernels_hash": bench.kernels_hash(), "entries": entries}
    with open(dst, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", dst, "hash", doc["kernels_hash"], "entries", list(entries))


if __name__ == "__main__":
    main()
