#!/usr/bin/env python3
"""usage: tools/update_traffic.py <workload> <summary.json from tools/prof.sh> <bench line json> <profiles/name.json>
Copies a PMC summary into profiles/ and records its HBM bytes per line in profiles/traffic.json together with the hash of
the sources the workload's kernel object was compiled from at measurement time (bench.py reports `roofline.traffic` only when
that hash matches the tree; summaries without per-workload hashes fall back to the hash over all kernel sources)."""
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
# the per-workload hash is computed HERE (the build directory with the compiler's dependency files does not travel to the GPU box),
# and only when this tree still is the one that was measured
sys.path.insert(0, str(root))
from flowgger_amd.build import source_hash  # noqa: E402

measured_all = s.get("src_hash")
# the summary carries the per-workload hashes as computed ON THE GPU BOX from the manifest that travels with the library
# (flowgger_amd/kernel_deps.json): that is the identity of what was measured, whatever has happened to this tree's host side since
measured_wl = (s.get("src_hashes") or {}).get(wl)
src_hash = measured_wl or (source_hash(wl) if measured_all and measured_all == source_hash() else measured_all)
tr = root / "profiles" / "traffic.json"
t = json.loads(tr.read_text()) if tr.exists() else {}
t[wl] = {"hbm_bytes_per_line": h["total"] / lines, "read": h["read"] / lines, "written": h["written"] / lines, "lines": lines,
         "profile": dest, "src_hash": src_hash, "src_hash_all": measured_all}
tr.write_text(json.dumps(t, indent=1) + "\n")
print(wl, t[wl])
