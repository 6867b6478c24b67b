#!/usr/bin/env python3
"""Registers / scratch / LDS / occupancy of every kernel of one translation unit: tools/kres.py fg_rfc5424.hip [-D...]
(hipcc -Rpass-analysis=kernel-resource-usage, cross-compiled for gfx950: no GPU needed).  The figure to watch after any kernel edit:
a VGPR count that crosses 128 / 168 / 256 changes the waves per SIMD, ScratchSize > 0 is spilling."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main():
    unit = sys.argv[1]
    extra = sys.argv[2:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-x", "hip", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off",
           f"-I{ROOT}/include", f"-I{ROOT}/tests/native", "-Rpass-analysis=kernel-resource-usage", "-c",
           str(ROOT / "flowgger_amd" / "csrc" / unit), "-o", "/dev/null"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-4000:])
        raise SystemExit(1)
    cur = None
    rows = {}
    for ln in r.stderr.splitlines():
        m = re.search(r"remark: +(?:Function )?Name: (\S+)", ln)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", cur)
            rows[cur] = {}
            continue
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[a-zA-Z/]+\])?: (\d+)", ln)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    for k, v in rows.items():
        print(f"{k:64s} VGPR {v.get('VGPRs', -1):4d} AGPR {v.get('AGPRs', -1):3d} SGPR {v.get('TotalSGPRs', v.get('SGPRs', -1)):4d} "
              f"scratch {v.get('ScratchSize', -1):5d} occ {v.get('Occupancy', -1):2d} LDS {v.get('LDS Size', -1):6d}")


main()
