"""CPU: `roofline.traffic` is reported only for the kernel sources it was measured on -- per workload: the files the workload's
kernel object was compiled from (compiler-written dependency files), so that an edit of one decoder does not disown the PMC
figure of another (flowgger_amd/build.py source_hash, bench.py, tools/update_traffic.py)."""
import json
from pathlib import Path

import pytest

from flowgger_amd import build as B

ROOT = Path(__file__).resolve().parent.parent


def _deps(unit):
    d = B._repo_deps(unit)
    if d is None:
        pytest.skip("no dependency files: the library was not built in this tree")
    return {f.name for f in d}


def test_a_workload_hash_covers_its_kernel_and_nothing_else():
    r = _deps("fg_rfc5424.hip")
    assert {"fg_rfc5424.hip", "fg_pipeline.hpp", "fg_wave.hpp", "fg_tsfast.hpp", "fg_timeconv.hpp", "fg_hip.h"} <= r
    assert not ({"fg_gelf2.hpp", "fg_rfc3164_parse.hpp", "fg_emit.hpp", "fg_numparse.hpp"} & r)
    g = _deps("fg_gelf.hip")
    assert {"fg_gelf.hip", "fg_gelf2.hpp", "fg_numfold.hpp", "fg_pipeline.hpp"} <= g and "fg_rfc3164_parse.hpp" not in g
    hashes = B.source_hashes()
    assert hashes["cfg2"] == hashes["cfg4"] == hashes["cfg5"]          # one kernel
    assert len({hashes["cfg2"], hashes["cfg3"], hashes["ltsv"], hashes["rfc3164"], hashes["*"]}) == 5
    assert B.source_hash("no such workload") == hashes["*"]


def test_host_side_edits_do_not_disown_a_kernel_figure(tmp_path, monkeypatch):
    """the host pipelines (copies, streams, events) were hashed into every kernel's identity, so host-side work reported
    four of five PMC traffic figures as null.  The host side of the C ABI is no unit of any workload now."""
    for w, units in B.WORKLOAD_UNITS.items():
        assert not ({"fg_capi.cpp", "fg_host_pipeline.cpp", "fg_ctx.hpp"} & set(units)), w
        for u in units:
            if u.endswith(".hip"):
                assert not ({"fg_capi.cpp", "fg_host_pipeline.cpp", "fg_ctx.hpp"} & _deps(u)), (w, u)
    # ... while the one host-picked tile size (RFC3164, encoders) still is part of those kernels' identity
    assert "fg_tile_cap.hpp" in B.WORKLOAD_UNITS["rfc3164"] and "fg_tile_cap.hpp" in B.WORKLOAD_UNITS["cfg1"]


def test_traffic_entries_are_stamped():
    """profiles/traffic.json: the entries bench.py may report carry the 16-hex-digit hash of the sources they were measured on
    (bench.py compares it with source_hash(workload) and reports null for anything else); entries without one are history."""
    import re

    t = json.loads((ROOT / "profiles" / "traffic.json").read_text())
    stamped = {w: v for w, v in t.items() if isinstance(v, dict) and v.get("src_hash")}
    assert {"cfg2", "cfg3"} <= set(stamped)
    for w, v in stamped.items():
        assert re.fullmatch(r"[0-9a-f]{16}", v["src_hash"]) and v["hbm_bytes_per_line"] > 0 and w in B.WORKLOAD_UNITS
