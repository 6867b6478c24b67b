#!/usr/bin/env python3
"""Static instruction counts between FG_MARK()s of one kernel (device assembly, -DFG_ASM_MARKS).

The decode kernels run at the SIMD's VALU issue rate (one wave64 VALU instruction per ~4 cycles), so for the straight-line phases
the number of v_* instructions between two marks is the phase's cost; loops show up once (multiply by the trip count yourself).
usage: tools/valu_count.py flowgger_amd/csrc/fg_gelf.hip k_gelfILi2ELb0ELi4E [-D...]"""
import re, subprocess, sys, os, collections

def main():
    src, pat = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = "/tmp/valu_count.s"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-x", "hip", "-O3", "-std=c++17", "-fno-fast-math", "-ffp-contract=off",
           "-DFG_ASM_MARKS", f"-I{root}/include", f"-I{root}/tests/native", "--cuda-device-only", "-S", src, "-o", out] + extra
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL) if not os.environ.get("VALU_REUSE") else None
    s = open(out).read()
    m = re.search(r"^(\S*%s[^\s:]*):.*\n" % re.escape(pat), s, re.M)
    if not m:
        sys.exit("kernel not found; candidates:\n" + "\n".join(re.findall(r"^(_Z[^\s:]+):.*$", s, re.M)))
    body = s[m.end():]
    body = body[:body.index(".end_amdhsa_kernel")] if ".end_amdhsa_kernel" in body else body
    print(m.group(1))
    kinds = ("v_", "s_", "ds_", "global_", "scratch_", "flat_", "buffer_")
    cur = collections.Counter(); total = collections.Counter(); label = "(entry)"
    rows = []
    for ln in body.split("\n"):
        t = ln.strip()
        if "FGMARK" in t:
            rows.append((label, cur)); cur = collections.Counter(); label = t.split("FGMARK")[1].strip(); continue
        if not t or t[0] in ";." or t.endswith(":"): continue
        op = t.split()[0]
        k = next((k for k in kinds if op.startswith(k)), "other")
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch")): k = "s_ctl"
        cur[k] += 1; total[k] += 1
    rows.append((label, cur))
    cols = ["v_", "s_", "s_ctl", "ds_", "global_", "scratch_", "flat_", "buffer_", "other"]
    print("%-10s" % "after" + "".join("%9s" % c for c in cols))
    for lab, c in rows:
        print("%-10s" % lab + "".join("%9d" % c[k] for k in cols))
    print("%-10s" % "total" + "".join("%9d" % total[k] for k in cols))
    for k in (".vgpr_count", ".sgpr_count", ".private_segment_fixed_size", ".vgpr_spill_count", ".sgpr_spill_count"):
        mm = re.search(r"\.name:\s+%s.*?%s:\s+(\d+)" % (re.escape(m.group(1)), re.escape(k)), s, re.S)
        # (metadata order varies: search backwards too)
        print(k, mm.group(1) if mm else "?")

main()
