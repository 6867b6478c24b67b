"""CPU: the C-ABI library loads, exports every symbol include/fg_hip.h declares, refuses to run
without a gfx950 GPU (no CPU fallback), and its pure-host entry points behave."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from flowgger_amd import _lib as L
from flowgger_amd.record import DecodeError, parse_canonical
from flowgger_amd.tables import HostTables, layout

ROOT = Path(__file__).resolve().parent.parent


def declared_functions():
    hdr = (ROOT / "include" / "fg_hip.h").read_text()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(fg_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    lib = L.lib()
    names = declared_functions()
    assert len(names) >= 12
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/fg_hip.h but not exported"
    assert lib.fg_abi_version() == 4


def test_product_never_touches_the_oracle():
    for p in list((ROOT / "flowgger_amd").rglob("*.py")) + list((ROOT / "flowgger_amd" / "csrc").glob("*")) + \
            list((ROOT / "include").glob("*")):
        if p.is_file() and p.suffix in (".py", ".cpp", ".hip", ".hpp", ".h"):
            txt = p.read_text(errors="replace")
            assert "fg_oracle" not in txt and "libfg_oracle" not in txt and "oracle_binding" not in txt, p


def test_product_library_reads_no_environment_variables():
    """tuning knobs out of the product launch path.  libfg_hip.so neither imports getenv nor carries the names of the
    old FG_* knobs; launch geometry is set per ctx (fg_set_launch_opts), the phase-clock kernels exist only in the measurement
    build (-DFG_PROF_BUILD -> libfg_hip_prof.so)."""
    import subprocess

    L.lib()
    blob = L.LIB_PATH.read_bytes()
    for knob in (b"FG_PROF", b"FG_PLAN", b"FG_ABLATE", b"FG_TILE_CAP", b"FG_LINES_PER_GROUP", b"FG_WAVES_PER_CU", b"FG_GELF_",
                 b"FG_TRANSCODE_ONE_PIECE"):
        assert knob not in blob, knob
    syms = subprocess.run(["nm", "-D", "--undefined-only", str(L.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms
    for src in (ROOT / "flowgger_amd" / "csrc").glob("*"):
        if src.suffix in (".hip", ".hpp", ".cpp"):
            txt = src.read_text(errors="replace")
            outside = re.sub(r"#if defined\(FG_PROF_BUILD\).*?#endif", "", txt, flags=re.S)
            assert "getenv" not in outside, f"{src.name}: getenv outside an FG_PROF_BUILD block"


def test_no_gpu_means_loud_failure():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    assert L.lib().fg_create(0, None, C.byref(ctx)) == L.FG_ERR_NO_DEVICE
    from flowgger_amd import RFC5424Decoder

    with pytest.raises(L.FgError):
        RFC5424Decoder()


def test_error_strings_match_oracle_strings(oracle):
    lib = L.lib()
    # every reachable RFC5424 error of the synthetic invalid tail maps to a table entry
    from flowgger_amd import synth

    table = {lib.fg_error_string(0, i).decode() for i in range(1, 18)}
    for line in synth.rfc5424_invalid_lines():
        res = parse_canonical(oracle.decode(0, line))
        assert isinstance(res, DecodeError) and str(res) in table, line
    assert lib.fg_error_string(0, 0) == b"" and lib.fg_error_string(0, 200) is None
    assert lib.fg_error_string(1, 9) == b"Unable to parse the English to Unix timestamp in LTSV decoder"
    assert lib.fg_error_string(2, 12) == b"Missing hostname"


def test_layout_and_shard_plan():
    offs, total = layout(1000, 500)
    assert [s for _, s in offs] == [4000, 8000] + [8000] * 6 + [4000, 4000, 4000, 4000, 500, 500, 8]
    assert all(o % 256 == 0 for o, _ in offs) and total >= sum(s for _, s in offs)
    lens = np.random.default_rng(1).integers(0, 700, 10000)
    offsets = np.zeros(10001, np.uint64)
    offsets[1:] = np.cumsum(lens)
    for g in (1, 2, 3, 8):
        starts = np.zeros(g + 1, np.uint64)
        assert L.lib().fg_shard_plan(offsets.ctypes.data, 10000, g, starts.ctypes.data) == 0
        assert starts[0] == 0 and starts[g] == 10000 and np.all(np.diff(starts.astype(np.int64)) >= 0)
        per = np.diff(offsets[starts.astype(np.int64)].astype(np.int64))
        assert per.max() - per.min() <= 2 * 700  # byte-balanced to within a line or two


def _host_tables(n, ents):
    offs, _ = layout(n, max(ents, 1))
    dt = {"meta": np.uint32, "ts": np.float64, "ent_val": np.uint64, "ent_type": np.uint8, "ent_flags": np.uint8,
          "ent_used": np.uint64}
    arrays = {}
    for name, (_, size) in zip(L.TABLE_FIELDS, offs):
        d = np.dtype(dt.get(name, np.uint32))
        arrays[name] = np.zeros(max(size // d.itemsize, 1), d)
    return HostTables(n, max(ents, 1), arrays)


def test_serialize_hand_built_rows(oracle):
    """fg_tables_serialize on rows written by hand == the oracle's canonical bytes for the line."""
    line = rb'<23>1 2015-08-05T15:53:45.637824Z testhostname appname 69 42 [origin@123 software="te\st sc\"ript" swVersion="0.0.1"] test message'
    data = np.frombuffer(line, np.uint8).copy()
    offsets = np.array([0, len(line)], np.uint64)
    t = _host_tables(1, 3)
    a = t.a
    a["meta"][0] = 0 | (2 << 8) | (7 << 16)
    a["ts"][0] = 1438790025.637824

    def span(col, s):
        o = line.index(s)
        a[col].reshape(-1, 2)[0] = (o, len(s))

    span("hostname", b"testhostname"); span("appname", b"appname"); span("procid", b"69"); span("msgid", b"42")
    span("msg", b"test message")
    a["full_msg"].reshape(-1, 2)[0] = (0, len(line))
    a["ent_first"][0], a["ent_count"][0] = 0, 3
    names = a["ent_name"].reshape(-1, 2)
    for k, (nm, val, ty, fl) in enumerate([(b"origin@123", None, 6, 0), (b"software", rb'te\st sc\"ript', 0, 1),
                                           (b"swVersion", b"0.0.1", 0, 0)]):
        names[k] = (line.index(nm), len(nm))
        if val is not None:
            a["ent_val"][k] = line.index(val) | (len(val) << 32)
        a["ent_type"][k], a["ent_flags"][k] = ty, fl
    a["ent_used"][0] = 3
    blob, offs = t.serialize(0, data, offsets)
    assert blob.tobytes() == oracle.decode(0, line)
    # an error row
    a["meta"][0] = 13
    blob, _ = t.serialize(0, data, offsets)
    assert str(parse_canonical(blob.tobytes())) == "Missing log message"


LTSV_NOVALUE_LINES = [
    b"time:1\thost:h\tnovalue\tmessage:m", b"novalue1\tnovalue2\ttime:1\thost:h", b"time:1\thost:h\t", b"\t\ttime:1\thost:h",
    b"a\tlevel:9\tb\ttime:1\thost:h", b"a\tb\tcounter:x\tc", b"time:1\thost:h\tlevel:3\tx y z\tscore:-4", b"nothing at all",
    b"", b"time:bad\tq", b"q\ttime:bad\tr", b"time:1\tq", b"k:v\tq\thost:h",
    "time:1\thost:h\tcaf\u00e9 \u4e2d\tmessage:ok".encode(),
]


def test_tables_stdout_reproduces_the_ltsv_println(oracle):
    """fg_tables_stdout (SURVEY 8b "Side effects", ltsv_decoder.rs:99): from rows flagged FG_F_LTSV_NOVALUE -- built here from the
    oracle's verdicts, by the kernels on the GPU (tests/test_gpu_round3.py) -- the exact text the reference prints, in order, and
    only for the parts it reached before a failing one."""
    from flowgger_amd import synth
    from flowgger_amd.tables import tables_stdout

    lines = LTSV_NOVALUE_LINES
    data, offsets = synth.pack(lines)
    data = np.concatenate([data, np.zeros(16, np.uint8)])
    t = _host_tables(len(lines), 1)
    want = b""
    for i, ln in enumerate(lines):
        text = oracle.decode_stdout(1, ln, synth.LTSV_CONFIG)
        res = parse_canonical(oracle.decode(1, ln, synth.LTSV_CONFIG))
        failed = isinstance(res, DecodeError)
        k = text.count(b"\n")
        t.a["meta"][i] = (7 if failed else 0) | (0xFF << 8) | (0xFF << 16) | ((128 if k else 0) << 24)
        t.a["hostname"].reshape(-1, 2)[i] = (k, 0xFFFFFFFF) if failed else (0, 1)
        want += text
    assert want.count(b"Missing value for name '") >= 12
    assert tables_stdout(t, 1, data, offsets) == want
    # frames that still carry their terminators
    for framing, term in ((1, b"\r\n"), (2, b"\0")):
        fdata, foffs = synth.pack([ln + term for ln in lines])
        fdata = np.concatenate([fdata, np.zeros(16, np.uint8)])
        assert tables_stdout(t, 1, fdata, foffs, framing=framing) == want
