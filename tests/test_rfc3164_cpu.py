"""CPU: the oracle's RFC3164 decoder against the reference's own decoder tests (rfc3164_decoder.rs:215-441), the zone
table builder against Python's zoneinfo, and the kernel's per-line parser (fg_rfc3164_parse.hpp, host build) against
the oracle on structured fuzz."""
import ctypes as C
import datetime
import random
import subprocess
import zoneinfo
from pathlib import Path

import numpy as np
import pytest

import oracle_binding as OB
from flowgger_amd import _lib as L
from flowgger_amd import synth, tzdb
from flowgger_amd.record import DecodeError, parse_canonical
from golden.reference_vectors import RFC3164, RFC3164_VECTORS, RFC3164_YEAR
from test_abi_cpu import _host_tables
from test_oracle_golden import check_vector

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def oracle3164(oracle):
    oracle.set_rfc3164(RFC3164_YEAR, tzdb.default_table())
    return oracle


@pytest.mark.parametrize("v", RFC3164_VECTORS, ids=[v["src"].split()[-1] for v in RFC3164_VECTORS])
def test_reference_vectors(oracle3164, v):
    check_vector(parse_canonical(oracle3164.decode(RFC3164, v["line"])), v)


def test_semantics(oracle3164):
    d = lambda s: parse_canonical(oracle3164.decode(RFC3164, s))  # noqa: E731
    # whitespace tokens are re-joined with single spaces (standard form); Unicode whitespace separates too
    r = d("Aug 6 11:15:24 host a\t b 　c  ")
    assert (r.hostname, r.msg, r.full_msg) == ("host", "a b c", "Aug 6 11:15:24 host a\t b 　c")
    assert d("Aug 6 11:15:24 host").msg == ""          # four tokens: hostname, empty message
    assert d("Aug 06 11:15:24 host m").ts == d("Aug 6 11:15:24 host m").ts   # [day padding:none] takes 1-2 digits
    # the date + zone consuming every token: the reference indexes out of bounds (:67)
    for s in ("Aug 6 11:15:24 UTC", "2020 Aug 6 11:15:24", "2020 Aug 6 11:15:24 Europe/Paris"):
        assert isinstance(d(s), DecodeError) and "panics" in str(d(s))
    # priority
    assert str(d("<13")) == "Malformed RFC3164 event: Invalid priority"
    assert str(d("<>Aug 6 11:15:24 h m")) == "Invalid priority" and str(d("<256>Aug 6 11:15:24 h m")) == "Invalid priority"
    r = d("<<+013>Aug 6 11:15:24 h m")
    assert (r.facility, r.severity) == (1, 5)
    assert (d("<255>Aug 6 11:15:24 h m").facility, d("<255>Aug 6 11:15:24 h m").severity) == (31, 7)
    # custom form errors surface (the standard form's never do)
    assert str(d("h: x: m")) == "Invalid time format"
    assert str(d("h: Aug 6 1:2:3: m")) == "Unable to parse RFC3164 date with year"
    assert str(d("h: 2020 Aug 6 1:2:3: m")) == "Unable to parse the date in RFC3164 decoder"
    assert str(d("h: Feb 30 11:15:24 x: m")) == "Unable to parse the date in RFC3164 decoder"
    r = d("my host: 2020 Feb 29 23:59:59: a: b: c ")
    assert (r.hostname, r.msg, r.ts) == ("my host", "a: b: c ", 1583020799.0)
    # month names are case-sensitive; a zone name is matched exactly
    assert isinstance(d("aug 6 11:15:24 host m"), DecodeError)
    assert d("Aug 6 11:15:24 utc host m").hostname == "utc"
    # zones: local -> UTC through the table; signed year
    assert d("2021 Jan 1 00:00:00 Asia/Tokyo h m").ts == 1609426800.0
    assert d("2021 Jul 1 12:00:00 America/New_York h m").ts == 1625155200.0
    assert d("-0001 Jan 1 00:00:00 h m").ts == float((datetime.date(1, 1, 1).toordinal() - 366 - 365 - 719163) * 86400)


def test_zone_table_equals_zoneinfo():
    T = tzdb.default_table()
    assert len(T.names) > 400 and T.names == sorted(T.names, key=lambda s: s.encode()) and "UTC" in T.names
    rnd = random.Random(7)
    for nm in rnd.sample(T.names, 60) + ["UTC", "America/Sao_Paulo", "Europe/Dublin", "Australia/Lord_Howe", "Africa/Casablanca"]:
        st, of = T.entries(nm)
        assert st[0] == tzdb.I64_MIN and np.all(np.diff(st.astype(np.float64)) > 0)
        z = zoneinfo.ZoneInfo(nm)
        for _ in range(100):
            u = rnd.randint(-2 ** 31, 4102444800 - 1)
            i = int(np.searchsorted(st, u, side="right")) - 1
            want = datetime.datetime.fromtimestamp(u, datetime.timezone.utc).astimezone(z).utcoffset().total_seconds()
            assert int(of[i]) == int(want), (nm, u)


def fuzz_lines(n, seed):
    r = random.Random(seed)
    zones = ["UTC", "America/Sao_Paulo", "Europe/Paris", "Asia/Kolkata", "Australia/Lord_Howe", "utc", "Nowhere/Land", "GMT", "EST5EDT"]
    mons = ["Jan", "Feb", "Mar", "Apr", "May", "Jun", "Jul", "Aug", "Sep", "Oct", "Nov", "Dec", "jan", "AUG", "Augu", "Xyz", ""]
    ws = [" ", " ", " ", "  ", "\t", " 　", " ", "  ", "\n", "​"]
    words = ["host", "app[12]:", "msg", "x:", ": ", "a: b", "é", "　", "42", "<1>", "UTC", "2020", "Aug", "11:15:24", "-", ""]
    out = []
    for _ in range(n):
        kind = r.random()
        pri = r.choice(["", "", "<13>", "<0>", "<191>", "<255>", "<256>", "<+7>", "<<5>", "<5>>", "<>", "<x>", "<13", "< 5>", "<-1>"])
        year = r.choice(["", "", "2020 ", "1999 ", "+2021 ", "-0044 ", "0000 ", "20200 ", "202 ", "9999 "])
        mon = r.choice(mons[:12]) if r.random() < 0.85 else r.choice(mons)
        day = r.choice([str(r.randint(1, 31)), "%02d" % r.randint(1, 28), "0", "32", "29", "30", "31", "x", "123"])
        tm = "%02d:%02d:%02d" % (r.randint(0, 23), r.randint(0, 59), r.randint(0, 59)) if r.random() < 0.85 else \
            r.choice(["24:00:00", "1:2:3", "11:15:60", "11:15", "11:15:24.5", "11-15-24", "23:59:59"])
        tz = r.choice(["", "", "", ""] + zones)
        sep = lambda: r.choice(ws)  # noqa: E731
        tail = sep().join(r.choice(words) for _ in range(r.randint(0, 6)))
        if kind < 0.55:      # standard
            s = pri + year + mon + sep() + day + sep() + tm + (sep() + tz if tz else "") + sep() + tail + r.choice(["", " ", "\n", " 　"])
        elif kind < 0.9:     # custom
            host = r.choice(["h", "my host", "", "h:", "a: b"])
            s = pri + host + ": " + year + mon + sep() + day + sep() + tm + (sep() + tz if tz else "") + r.choice([": ", ":", " : ", ": : "]) + tail
        else:
            s = pri + sep().join(r.choice(words + mons) for _ in range(r.randint(0, 8)))
        out.append(s.encode("utf-8"))
    return out


@pytest.fixture(scope="module")
def host3164():
    src, lib = ROOT / "tests/native/rfc3164_host.cpp", ROOT / "tests/native/librfc3164_host.so"
    deps = [src, ROOT / "flowgger_amd/csrc/fg_rfc3164_parse.hpp", ROOT / "flowgger_amd/csrc/fg_tz_index.hpp", ROOT / "flowgger_amd/csrc/fg_timeconv.hpp", ROOT / "include/fg_hip.h"]
    if not lib.exists() or lib.stat().st_mtime < max(p.stat().st_mtime for p in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math", "-o", str(lib), str(src)], check=True)
    lib = C.CDLL(str(lib))
    T = tzdb.default_table()
    names = (C.c_char_p * len(T.names))(*[n.encode() for n in T.names])
    zf, us, uo = T.zone_first.astype(np.uint32), T.utc_start.astype(np.int64), T.utc_offset.astype(np.int32)

    def run(lines, year=RFC3164_YEAR):
        data, offsets = synth.pack(lines)
        t = _host_tables(len(lines), 1)
        lib.fg3_decode_batch(C.c_void_p(data.ctypes.data), C.c_void_p(offsets.ctypes.data), C.c_uint64(len(lines)), C.c_int32(year),
                             C.c_uint32(len(T.names)), names, C.c_void_p(zf.ctypes.data), C.c_void_p(us.ctypes.data),
                             C.c_void_p(uo.ctypes.data), C.byref(t.struct))
        return t.serialize(L.FG_RFC3164, data, offsets), data, offsets
    return run


def test_kernel_parser_equals_oracle(host3164, oracle3164):
    """fg_rfc3164_parse.hpp (host build of what the kernel runs) + the product's materialiser == the oracle, byte for
    byte, on the reference's vectors, the synthetic corpus and structured fuzz (all error paths)."""
    lines = [v["line"].encode() for v in RFC3164_VECTORS] + synth.rfc3164_lines(4000) + fuzz_lines(20000, 11)
    (blob, offs), data, offsets = host3164(lines)
    oblob, ooffs = oracle3164.decode_batch(RFC3164, data, offsets)
    for i in range(len(lines)):
        a, b = blob[int(offs[i]):int(offs[i + 1])].tobytes(), oblob[int(ooffs[i]):int(ooffs[i + 1])].tobytes()
        assert a == b, (i, lines[i], parse_canonical(a), parse_canonical(b))
    res = [parse_canonical(oblob[int(ooffs[i]):int(ooffs[i + 1])].tobytes()) for i in range(len(lines))]
    errs = {str(r) for r in res if isinstance(r, DecodeError)}
    assert len(errs) == 7, errs                      # every status is exercised
    assert sum(not isinstance(r, DecodeError) for r in res) > 0.2 * len(lines)


def test_zone_index_near_misses_and_unhinted_years(host3164, oracle3164):
    """the hash index / reject masks / year hints of fg_tz_index.hpp: tokens that pass the masks but are no zone,
    every zone name of the table, and dates (with a year) far outside the hinted year -- all == the oracle's
    linear get_by_name + assume_timezone"""
    T = tzdb.default_table()
    names = list(T.names)
    lines = []
    for k, z in enumerate(names):  # every zone, in the configured year (hinted) -- both halves of the year (DST)
        mon = ("Jan", "Jul", "Mar", "Oct", "Nov")[k % 5]
        lines.append(f"<13>{mon} {1 + k % 28} 0{k % 10}:30:1{k % 10} {z} host{k} message {k}".encode())
    for k, z in enumerate(names[::7]):  # with a year: far past, far future, around the first / last transitions
        for year in (1850, 1901, 1916, 1942, 1970, 1986, 2007, 2037, 2038, 2100, 9999):
            lines.append(f"{year} Mar {25 + k % 6} 02:30:00 {z} h m".encode())
            lines.append(f"{year} Oct {25 + k % 6} 02:30:00 {z} h m".encode())
    near = []
    for z in names[::11]:  # near misses: same first byte and a plausible length, not a zone
        near += [z[:-1], z + "s", z.lower(), z.upper(), z[0] + z[1:].swapcase(), z.replace("/", "_"), z + "/", "/" + z]
    near += ["UTCx", "UT", "U", "Z", "Zulu0", "EST5ED", "GMT+", "GMT-15", "Etc/GMT+13", "É", "Ünicode", "Europe", "Europe/"]
    known = set(names)
    for k, tok in enumerate(near):
        if tok and tok not in known and " " not in tok:
            lines.append(f"Aug  6 11:15:24 {tok} rest of the message {k}".encode())  # tok is the hostname
            lines.append(f"Aug  6 11:15:24 {tok}".encode())                          # ... and nothing follows
    (blob, offs), data, offsets = host3164(lines)
    oblob, ooffs = oracle3164.decode_batch(RFC3164, data, offsets)
    for i in range(len(lines)):
        a, b = blob[int(offs[i]):int(offs[i + 1])].tobytes(), oblob[int(ooffs[i]):int(ooffs[i + 1])].tobytes()
        assert a == b, (i, lines[i], parse_canonical(a), parse_canonical(b))
    # a different configured year moves the hint window: same equality
    (blob, offs), data, offsets = host3164(lines[:len(names)], year=1999)
    oracle3164.set_rfc3164(1999, T)
    try:
        oblob, ooffs = oracle3164.decode_batch(RFC3164, data, offsets)
    finally:
        oracle3164.set_rfc3164(RFC3164_YEAR, T)
    assert np.array_equal(offs, ooffs) and np.array_equal(blob, oblob)


def test_sixteen_byte_scans_at_every_offset(host3164, oracle3164):
    """The token scan and the custom form's ": " search move sixteen bytes per step: whitespace (ASCII and multi-byte), ':' and
    ": " at every offset of the windows, tokens of 1..48 bytes, non-ASCII bytes that are NOT whitespace inside tokens."""
    r = random.Random(316416)
    ws = [" ", "\t", "\n", "\u00a0", "\u2003", "\u3000", "\u0085", "  "]
    filler = "abcdefghijklmnopqrstuvwxyz0123456789-_.:/[]\u00e9\u4e2d\u200b"   # (U+200B is NOT White_Space)
    lines = []
    for k in range(6000):
        def tok(n):
            return "".join(r.choice(filler) for _ in range(n))
        date = "Aug %2d 11:15:24" % r.randint(1, 28)
        host = tok(r.randint(1, 48)).replace(": ", ":_")
        sep = r.choice(ws) if r.random() < 0.3 else " "
        pri = r.choice(["", "<13>", "<191>"])
        if k % 3 == 0:   # standard form, long host / first message token
            lines.append(f"{pri}{date}{sep}{host}{sep}{tok(r.randint(1, 40))} {tok(r.randint(0, 30))}")
        elif k % 3 == 1:  # custom form: the two separators anywhere
            lines.append(f"{pri}{host}: 2020 {date}: {tok(r.randint(0, 60))}")
        else:             # separators / colons at the edges of the windows
            pad = tok(r.randint(0, 40)).replace(": ", "::")
            lines.append(f"{pri}{pad}:{r.choice(['', ' ', '  '])}{date}{r.choice([':', ': ', ' :', ''])}{tok(r.randint(0, 20))}{r.choice(['', ':', ': '])}")
    enc = [ln.encode("utf-8") for ln in lines]
    (blob, offs), data, offsets = host3164(enc)
    oblob, ooffs = oracle3164.decode_batch(RFC3164, data, offsets)
    for i in range(len(enc)):
        a, b = blob[int(offs[i]):int(offs[i + 1])].tobytes(), oblob[int(ooffs[i]):int(ooffs[i + 1])].tobytes()
        assert a == b, (i, enc[i], parse_canonical(a), parse_canonical(b))
