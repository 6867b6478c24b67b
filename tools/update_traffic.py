#!/usr/bin/env python3
"""usage: tools/update_traffic.py <workload> <summary.json from tools/prof.sh> <bench line json> <profiles/name.json>
Copies a PMC summary into profiles/ and records its HBM bytes per line in profiles/traffic.json together with the hash of
the kernel sources it was measured on (bench.py reports `roofline.traffic` only when that hash matches the tree)."""
import json
import shutil
import sys
from pathlib import Path

wl, summary, bench_line, dest = sys.argv[1:5]
root = Path(__file__).resolve().parent.parent
s = json.load(open(summary))
b = json.loads(open(bench_line).read().strip().splitlines()[-1])
lines = b["config"]["lines_per_gpu"]
h = s["hbm_bytes_per_dispatch"]
shutil.copy(summary, root / dest)
tr = root / "profiles" / "traffic.json"
t = json.loads(tr.read_text()) if tr.exists() else {}
t[wl] = {"hbm_bytes_per_line": h["total"] / lines, "read": h["read"] / lines, "written": h["written"] / lines, "lines": lines,
         "profile": dest, "src_hash": s.get("src_hash")}
tr.write_text(json.dumps(t, indent=1) + "\n")
print(wl, t[wl])
