"""CPU: the encoder EMITTERS the kernels run (flowgger_amd/csrc/fg_emit.hpp + fg_enc_cfg.hpp, host build in
tests/native/emit_host.cpp) against the oracle's encoders, record by record: a canonical Record is laid out as a
synthetic source line + table row (raw / RFC5424-SD-escaped / JSON-escaped spans with the matching flags), emitted
through every encoder x merger, and compared with oracle.encode() of the same Record."""
import ctypes as C
import random
import struct
import subprocess
from pathlib import Path

import pytest

import oracle_binding as OB
from test_encoder_cpu import canonical
from test_encoders_cpu import LTSV_VECTORS, RFC3164_VECTORS, RFC5424_VECTORS

ROOT = Path(__file__).resolve().parent.parent
RFC5424, LTSV, GELF = 0, 1, 2
ES = {0: None, 2: "Failed to parse date", 3: "Failed to parse date as Rfc3339 format",
      4: "Failed to parse unix timestamp in RFC3164 encoder", 5: "Cannot output empty raw message"}


@pytest.fixture(scope="module")
def emit():
    src, lib = ROOT / "tests/native/emit_host.cpp", ROOT / "tests/native/libemit_host.so"
    deps = [src] + [ROOT / "flowgger_amd/csrc" / n for n in ("fg_emit.hpp", "fg_enc_cfg.hpp", "fg_shortest.hpp", "fg_dtoa.hpp",
                                                             "fg_tables_view.hpp", "fg_timeconv.hpp", "fg_unicode_ws.hpp")] + [ROOT / "include/fg_hip.h"]
    if not lib.exists() or lib.stat().st_mtime < max(p.stat().st_mtime for p in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math", "-Wno-unknown-pragmas", "-o", str(lib), str(src)],
                       check=True)
    L = C.CDLL(str(lib))
    L.fge_encode_canonical.restype = C.c_int64
    L.fge_encode_canonical.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                       C.c_uint32, C.c_char_p, C.c_double, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint32)]

    def run(enc, merger, src_fmt, rec: bytes, extra=None, prepend=None, now_ts=0.0, variant=0, seed=1):
        items = sorted((extra or {}).items())
        ks = (C.c_char_p * max(len(items), 1))(*[k.encode() for k, _ in items])
        vs = (C.c_char_p * max(len(items), 1))(*[v.encode() for _, v in items])
        st = C.c_uint32()
        buf = C.create_string_buffer(len(rec) * 8 + 4096)
        n = L.fge_encode_canonical(enc, merger, src_fmt, variant, seed, rec, len(rec), ks, vs, len(items),
                                   None if prepend is None else prepend.encode(), now_ts, buf, len(buf), C.byref(st))
        assert n >= 0, n
        assert n <= len(buf)
        return (buf.raw[:n] if st.value == 0 else ES[st.value])
    run.set_sort_slots = L.fge_set_sort_slots
    L.fge_last_plain.restype = C.c_uint32
    run.last_plain = L.fge_last_plain
    L.fge_sink_fuzz.argtypes = [C.c_uint64, C.c_uint32]
    run.sink_fuzz = L.fge_sink_fuzz
    return run


def norm_keys(rec):
    """Record keys as the decoders produce them (always a leading '_')."""
    rec = dict(rec)
    if rec.get("sd"):
        rec["sd"] = [(i, [(k if k.startswith("_") else "_" + k, v) for k, v in pairs]) for i, pairs in rec["sd"]]
    return rec


def test_reference_vectors_through_the_emitters(emit, oracle):
    for v in RFC5424_VECTORS:
        assert emit(OB.ENC_RFC5424, 0, RFC5424, canonical(**norm_keys(v[1]))).decode() == v[2]
    for v in LTSV_VECTORS:
        want = v[2].format(ts=oracle.rust_display(v[1]["ts"]))
        assert emit(OB.ENC_LTSV, 0, RFC5424, canonical(**norm_keys(v[1]))).decode() == want
    for v in RFC3164_VECTORS:
        assert emit(OB.ENC_RFC3164, 0, RFC5424, canonical(**norm_keys(v[1])), prepend=v[2]).decode() == v[3]
    rec = canonical(ts=1.2, hostname="abcd", msg="test message", full_msg="raw")
    assert emit(OB.ENC_PASSTHROUGH, 0, RFC5424, rec) == b"raw"
    assert emit(OB.ENC_PASSTHROUGH, 0, RFC5424, canonical(ts=1.2, hostname="abcd")) == "Cannot output empty raw message"
    # gelf_encoder.rs:124-150 with the keys as a decoder would have produced them
    g = dict(ts=1385053862.3072, hostname="example.org", severity=1, appname="appname", procid="44", msg="m",
             full_msg="Backtrace here\n\nmore stuff", sd=[("someid", [("_some_info", (0, "foo"))])])
    assert emit(OB.ENC_GELF, 0, RFC5424, canonical(**g), extra={"secret-token": "secret"}) == \
        oracle.encode(OB.ENC_GELF, canonical(**g), extra={"secret-token": "secret"})


ALPHA = ["a", "b", "Z", "0", "_", "-", ".", " ", "\t", "\n", ":", '"', "\\", "]", "[", "=", "/", "\x01", "\x1f", "\x7f", "\r", "\x08", "\x0c",
         "é", "ß", "€", "　", "𝄞", "😀", "n", "u", "t"]


def rstr(r, lo=0, hi=12):
    return "".join(r.choice(ALPHA) for _ in range(r.randint(lo, hi)))


def rkey(r):
    # small alphabet: duplicates, shared 7-byte prefixes, prefixes of the static GELF keys
    base = r.choice(["", "k", "key", "longprefix", "longprefiy", "host", "_x", "sd_id", "é", "a:b\tc", "ver\\sion", 'q"'])
    return "_" + base + ("" if r.random() < 0.3 else r.choice(["", "1", "2", "_bool", "é", rstr(r, 0, 4)]))


def rvalue(r, src):
    ty = r.choice([0, 0, 0, 1, 2, 3, 4, 5]) if src != RFC5424 else 0
    if ty == 0:
        return (0, rstr(r))
    if ty == 1:
        return (1, r.random() < 0.5)
    if ty == 2:
        return (2, r.choice([0.0, -0.0, 1.5, 123.456, 1e21, 1e-7, float("nan"), float("inf"), float("-inf"), r.random() * 10 ** r.randint(-10, 25),
                             struct.unpack("<d", struct.pack("<Q", r.getrandbits(64)))[0]]))
    if ty == 3:
        return (3, r.choice([0, -1, -2 ** 63, 2 ** 63 - 1, r.randint(-10 ** 12, 10 ** 12)]))
    if ty == 4:
        return (4, r.choice([0, 2 ** 64 - 1, r.randint(0, 10 ** 15)]))
    return (5, None)


def rrecord(r, src):
    opt = lambda f: f() if r.random() < 0.7 else None
    ts = r.choice([0.0, -0.5, 1438790025.637824, 1438859724.0, 253402300799.9, 253402300800.0, -62167219200.5, -62167219201.0,
                   -377705116800.0, -377705116801.0, 1e25, -1e25, 1e300, float("nan"), float("inf"), float("-inf"), 9.3e18, -9.3e18,
                   r.uniform(-4e11, 3e11), r.randint(0, 2 ** 32) + r.randint(0, 999) / 1000.0,
                   struct.unpack("<d", struct.pack("<Q", r.getrandbits(64)))[0]])
    sd = None
    if r.random() < 0.75:
        if src == RFC5424:
            sd = [(rstr(r, 0, 6), [(rkey(r), rvalue(r, src)) for _ in range(r.choice([0, 1, 2, 3, 5, 12, 33, 40]))]) for _ in range(r.randint(1, 3))]
        else:
            sd = [(None, [(rkey(r), rvalue(r, src)) for _ in range(r.choice([1, 2, 3, 5, 12, 33, 40]))])]
    return dict(ts=ts, hostname=rstr(r, 0, 8), facility=opt(lambda: r.randint(0, 31)), severity=opt(lambda: r.randint(0, 7)),
                appname=opt(lambda: rstr(r, 0, 6)), procid=opt(lambda: rstr(r, 0, 5)), msgid=opt(lambda: rstr(r, 0, 5)),
                msg=opt(lambda: rstr(r, 0, 30)), full_msg=opt(lambda: rstr(r, 0, 40)), sd=sd)


@pytest.mark.parametrize("src", [RFC5424, LTSV, GELF], ids=["src_rfc5424", "src_ltsv", "src_gelf"])
@pytest.mark.parametrize("enc", [OB.ENC_GELF, OB.ENC_LTSV, OB.ENC_RFC5424, OB.ENC_RFC3164, OB.ENC_PASSTHROUGH],
                         ids=["gelf", "ltsv", "rfc5424", "rfc3164", "passthrough"])
def test_emitters_equal_oracle_on_random_records(emit, oracle, src, enc):
    r = random.Random(1000 * src + enc)
    extras = [None, {"secret-token": "secret"}, {"_k": "shadow", "host": "h2", "_zz\tq:": 'v"\t\n\\', "a": ""}]
    n_ok = 0
    for i in range(700):
        rec = rrecord(r, src)
        cb = canonical(**rec)
        merger = r.randint(0, 3)
        extra = r.choice(extras)
        prepend = r.choice([None, None, "", "2026-09-23T10:11Z "])
        got = emit(enc, merger, src, cb, extra=extra, prepend=prepend, now_ts=12.5, variant=i & 1, seed=i)
        want = oracle.encode(enc, cb, merger, extra=extra, prepend=prepend, now_ts=12.5)
        assert got == want, (i, rec, merger, extra, prepend)
        n_ok += isinstance(want, bytes)
    assert n_ok > 100


def test_gelf_emitter_with_the_small_ranking_scratch(emit, oracle):
    """the kernels pick an 8-entry ranking scratch for batches with few pairs per line: lines with more pairs take the
    exact selection path and must give the same bytes"""
    emit.set_sort_slots(8)
    try:
        r = random.Random(77)
        for i in range(600):
            src = r.choice([RFC5424, LTSV, GELF])
            rec = rrecord(r, src)
            cb = canonical(**rec)
            extra = r.choice([None, {"_k": "shadow", "host": "h2"}])
            assert emit(OB.ENC_GELF, 1, src, cb, extra=extra, variant=i & 1, seed=i) == oracle.encode(OB.ENC_GELF, cb, 1, extra=extra), (i, rec)
    finally:
        emit.set_sort_slots(32)


def test_ts_now_uses_the_callers_clock(emit, oracle):
    # canonical ts_kind = 1 (GELF record without "timestamp")
    cb = bytearray(canonical(ts=0.0, hostname="h", msg="m"))
    cb[1] = 1
    cb = bytes(cb)
    for enc in (OB.ENC_GELF, OB.ENC_LTSV, OB.ENC_RFC5424, OB.ENC_RFC3164):
        got = emit(enc, 1, GELF, cb, now_ts=1438859724.638)
        assert got == oracle.encode(enc, cb, 1, now_ts=1438859724.638)
        assert b"1438859724.638" in got or b"11:15:24" in got


def test_integer_text_at_the_chunk_boundaries(emit, oracle):
    """u64_digits prints 2 + 9 + 9 digit chunks with the leading zeros suppressed: every power-of-ten boundary, the
    chunk borders (10^9, 10^18) and the type limits, as U64 / I64 pair values through the LTSV and GELF encoders and as
    the syslen prefix (message lengths around 10, 100, 1000)"""
    vals = sorted({0, 2 ** 64 - 1, 2 ** 63, 2 ** 63 - 1} | {10 ** k + d for k in range(0, 20) for d in (-1, 0, 1) if 0 <= 10 ** k + d < 2 ** 64})
    for src in (LTSV, GELF):
        for i in range(0, len(vals), 8):
            pairs = [(f"_u{j}", (4, v)) for j, v in enumerate(vals[i:i + 8])]
            pairs += [(f"_i{j}", (3, -v if v <= 2 ** 63 else -(2 ** 63))) for j, v in enumerate(vals[i:i + 8])]
            cb = canonical(ts=1.5, hostname="h", severity=5, facility=23, msg="m", sd=[(None, pairs)])
            for enc in (OB.ENC_LTSV, OB.ENC_GELF, OB.ENC_RFC5424):
                assert emit(enc, 0, src, cb) == oracle.encode(enc, cb, 0), (src, enc, vals[i:i + 8])
    for n in (1, 5, 6, 7, 95, 96, 97, 98, 995, 996, 997, 998, 9996, 9997, 9998):  # framed length crosses 10 / 100 / 1000 / 10000
        cb = canonical(ts=1.5, hostname="h", msg="m", full_msg="x" * n)
        assert emit(OB.ENC_PASSTHROUGH, 3, RFC5424, cb) == oracle.encode(OB.ENC_PASSTHROUGH, cb, 3), n


@pytest.mark.parametrize("enc", [OB.ENC_GELF, OB.ENC_LTSV, OB.ENC_RFC5424, OB.ENC_RFC3164, OB.ENC_PASSTHROUGH],
                         ids=["gelf", "ltsv", "rfc5424", "rfc3164", "passthrough"])
def test_long_spans_every_tail_length_and_the_plain_note(emit, oracle, enc):
    """Spans of 0 .. 70 bytes at every length (full 16-byte steps + every tail of 0 .. 15 bytes: the tail is ONE piece, a string's
    closing quote rides on it), plain and with a byte to escape at the start, at the end, in the tail and across a 16-byte boundary;
    timestamps that are assembled in registers (seconds with and without a fraction) and ones that stream.  The count pass's note
    for the write pass -- no span holds a byte to escape -- must be set for the plain records (the write pass then copies untested;
    the harness runs it both ways) and clear whenever the GELF encoder had to escape."""
    r = random.Random(77 + enc)
    text = "The quick brown fox jumps over the lazy dog 0123456789 +-*/=()<>{}|~^%$#@!?;,.'`&_"
    seen = [0, 0]
    for n in range(0, 71):
        for kind in range(6):
            body = (text * 2)[n % 7: n % 7 + n]
            if kind and n:
                at = {1: 0, 2: n - 1, 3: max(0, n - 1 - (n % 16) // 2), 4: min(n - 1, 15), 5: r.randrange(n)}[kind]
                body = body[:at] + r.choice(['"', "\\", "\n", "\t", "\x01", "\x1f"]) + body[at + 1:]
            ts = [1438790025.637824, 1438859724.0, 0.5, 1e15, 1234567.125, 1e22, 1e-9][(n + kind) % 7]
            rec = dict(ts=ts, hostname=body[: 1 + n % 40] or "h", severity=n % 8, facility=n % 24, appname="app" if n % 3 else None,
                       procid=str(n) if n % 4 else None, msgid="ID" if n % 5 else None, msg=body if n % 11 else None,
                       full_msg=("<13>1 - " + body) if n % 13 else None)
            cb = canonical(**rec)
            for merger in (0, 1, 3):
                got = emit(enc, merger, RFC5424, cb, now_ts=1.5)
                want = oracle.encode(enc, cb, merger, now_ts=1.5)
                assert got == want, (n, kind, merger, rec)
                if isinstance(want, bytes) and enc == OB.ENC_GELF:
                    escaped = b"\\" in want
                    seen[emit.last_plain()] += 1
                    assert emit.last_plain() == (0 if escaped else 1), (n, kind, rec)
    if enc == OB.ENC_GELF:
        assert seen[0] > 100 and seen[1] > 100


def test_pack_sink_alone_any_sequence_of_pieces(emit):
    """PackSink (the write pass's sink) on its own: random sequences of one-byte puts, 1-4-byte words, sixteen-byte copies and pieces of
    1-16 bytes at random start alignments leave exactly the bytes they were handed -- nothing before, nothing behind (guard bytes)."""
    for seed in range(400):
        assert emit.sink_fuzz(seed, 1 + seed % 97) == 0, seed
    for seed in range(40):
        assert emit.sink_fuzz(10_000 + seed, 5000) == 0, seed
