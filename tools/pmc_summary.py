#!/usr/bin/env python
"""Aggregates rocprofv3 --pmc CSV output (one *_counter_collection.csv per pass) into per-kernel, per-launch means:
   python tools/pmc_summary.py <out.json> <csv> [<csv> ...]
Kernel names are shortened to the template head (e.g. conv_halo_kernel<7, 1, 8>).  FETCH_SIZE/WRITE_SIZE are in KB
(rocprofv3 derived counters); hbm_bytes applies the gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE
counts 128-B requests at 64 B, so reads are doubled; WRITE_SIZE is taken as is (uncalibrated)."""
import csv
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_id():
    """sha256 (first 16 hex digits) over the two built libraries the counters were collected on: bench.py quotes a PMC file only when it
    describes the build it is running (kg_instance_segmentation_amd/libkgnet_hip*.so)"""
    h = hashlib.sha256()
    for n in ("libkgnet_hip.so", "libkgnet_hip_f16.so"):
        with open(os.environ.get("KG_LIB_PATH" if n == "libkgnet_hip.so" else "KG_LIB_F16_PATH") or os.path.join(ROOT, "kg_instance_segmentation_amd", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]

csv.field_size_limit(1 << 30)


def short(name):
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[^(]*?>)?)\(", name)
    s = m.group(1) if m else name
    return s if len(s) < 120 else s[:117] + "..."


def main():
    out = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in sys.argv[2:]:
        per_dispatch = defaultdict(float)
        names = {}
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        for (disp, ctr), v in per_dispatch.items():
            a = acc[names[disp]][ctr]
            a[0] += v; a[1] += 1
    res = {}
    for k, ctrs in acc.items():
        d = {c: v[0] / v[1] for c, v in ctrs.items()}
        d["launches"] = max(v[1] for v in ctrs.values())
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        res[k] = d
    res = dict(sorted(res.items(), key=lambda kv: -kv[1].get("hbm_bytes", 0) * kv[1]["launches"]))
    res["_build"] = {"libs_sha256_16": build_id(), "note": "hash of libkgnet_hip.so + libkgnet_hip_f16.so the counters were collected on"}
    json.dump(res, open(out, "w"), indent=1)
    for k, d in [kv for kv in res.items() if kv[0] != "_build"][:12]:
        print(f"{k[:60]:60s} n={d['launches']:4d} hbm/launch={d.get('hbm_bytes', 0) / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
