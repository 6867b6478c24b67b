#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace stats + PMC counter collection) into a small text
summary that can be committed under profiles/.  usage: prof_summary.py <dir> [kernel-substring]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_"
out = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    out.setdefault("kernel_stats", []).extend(
        {k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs") if k in r}
        for r in rows[:12])
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: [0.0, 0])
    extra = {}
    for r in csv.DictReader(open(f)):
        if pat not in r.get("Kernel_Name", ""):
            continue
        acc[r["Counter_Name"]][0] += float(r["Counter_Value"])
        acc[r["Counter_Name"]][1] += 1
        for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size"):
            if k in r:
                extra[k] = r[k]
    out.setdefault("pmc_per_dispatch_mean", {}).update({k: v[0] / max(v[1], 1) for k, v in acc.items()})
    out.setdefault("dispatch_info", {}).update(extra)
pm = out.get("pmc_per_dispatch_mean", {})
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the
    # bytes of a wide coalesced streaming read -> doubled.  WRITE_SIZE matched the 68 B/line row stores exactly in
    # r01b, so it is taken as is.
    out["hbm_bytes_per_dispatch"] = {"read": 2 * pm["FETCH_SIZE"] * 1024, "written": pm["WRITE_SIZE"] * 1024,
                                     "total": 2 * pm["FETCH_SIZE"] * 1024 + pm["WRITE_SIZE"] * 1024}
try:  # which kernel sources these numbers belong to (bench.py reports a traffic figure only for matching sources)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from flowgger_amd.build import source_hash, source_hashes

    out["src_hash"] = source_hash()
    out["src_hashes"] = source_hashes()  # per workload: the files its kernel object was compiled from
except Exception as e:  # pragma: no cover
    out["src_hash"] = None
    out["src_hashes"] = {}
print(json.dumps(out, indent=1))
