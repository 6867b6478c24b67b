"""CPU: the pair-parallel structured-data walk of the RFC5424 kernel (flowgger_amd/csrc/fg_sd2.hpp) -- the very source the kernel
compiles, built by g++ over the fiber emulation of a wavefront (tests/native/sd2_host.cpp) -- against the reference's state machine
(rfc5424_decoder.rs:127-242) run byte by byte over the same lines: every line the fast form HANDLES must give the reference's
status, message start and entries, entry by entry; what it hands back is only counted (the kernel walks those byte-wise)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

from flowgger_amd import synth

HERE = Path(__file__).resolve().parent / "native"
ROOT = HERE.parent.parent
LIB = HERE / "libsd2_host.so"
SRC = [HERE / "sd2_host.cpp", HERE / "fg_wave_emu.hpp", ROOT / "flowgger_amd/csrc/fg_sd2.hpp", ROOT / "flowgger_amd/csrc/fg_wave.hpp",
       ROOT / "flowgger_amd/csrc/fg_tables_view.hpp", ROOT / "include/fg_hip.h"]


@pytest.fixture(scope="module")
def lib():
    if not LIB.exists() or any(s.stat().st_mtime > LIB.stat().st_mtime for s in SRC):
        subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas",
                        f"-I{ROOT / 'include'}", f"-I{HERE}", f"-I{ROOT / 'flowgger_amd' / 'csrc'}", "-o", str(LIB), str(HERE / "sd2_host.cpp")],
                       check=True)
    L = C.CDLL(str(LIB))
    L.fgs2_last_error.restype = C.c_char_p
    L.fgs2_walk.restype = C.c_long
    return L


def sd_pos_of(ln: bytes) -> int:
    """line index of part 7 when it starts with '[' (splitn(7, ' ')), else 0"""
    p = 0
    for _ in range(6):
        q = ln.find(b" ", p)
        if q < 0:
            return 0
        p = q + 1
    return p if p < len(ln) and ln[p:p + 1] == b"[" and p >= 32 else 0


def walk(lib, lines, lines_per_group=64, tile_cap=18432, head_cap=0):
    data, offsets = synth.pack(lines)
    n = len(lines)
    data = np.concatenate([data, np.zeros(64, np.uint8)])
    sd_pos = np.array([sd_pos_of(ln) for ln in lines], np.uint32)
    cap = int(data.size) // 4 + 1024
    u32 = lambda k: np.zeros(max(k, 1), np.uint32)  # noqa: E731
    handled, bailed = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
    status, msg_at, n_ent, first, ent = u32(n), u32(n), u32(n), u32(n), u32(6 * cap)
    rstatus, rmsg, rfirst, rn, rent = u32(n), u32(n), u32(n), u32(n), u32(6 * cap)
    p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    rc = lib.fgs2_walk(p(data), C.c_uint64(data.size - 64), p(offsets), C.c_uint64(n), p(sd_pos), C.c_uint32(lines_per_group), C.c_uint32(tile_cap),
                       C.c_uint32(head_cap), p(handled), p(status), p(msg_at), p(n_ent), p(first), p(ent), C.c_uint64(cap), p(rstatus), p(rmsg),
                       p(rfirst), p(rn), p(rent), p(bailed))
    assert rc >= 0, lib.fgs2_last_error().decode()
    got = dict(handled=handled[:n], status=status[:n], msg_at=msg_at[:n], n_ent=n_ent[:n], first=first[:n], ent=ent.reshape(-1, 6), sd_pos=sd_pos,
               bailed=bailed[:n])
    ref = dict(status=rstatus[:n], msg_at=rmsg[:n], n_ent=rn[:n], first=rfirst[:n], ent=rent.reshape(-1, 6))
    return got, ref


def check(lib, lines, min_handled=None, **geom):
    got, ref = walk(lib, lines, **geom)
    n_sd = int((got["sd_pos"] != 0).sum())
    n_handled = 0
    for i, ln in enumerate(lines):
        if not got["sd_pos"][i]:
            assert not got["handled"][i]
            continue
        if not got["handled"][i]:
            continue
        n_handled += 1
        assert got["status"][i] == ref["status"][i], (i, ln, int(got["status"][i]), int(ref["status"][i]))
        if ref["status"][i] == 0:
            assert got["msg_at"][i] == ref["msg_at"][i], (i, ln)
            assert got["n_ent"][i] == ref["n_ent"][i], (i, ln, int(got["n_ent"][i]), int(ref["n_ent"][i]))
            a = got["ent"][got["first"][i]:got["first"][i] + got["n_ent"][i]]
            b = ref["ent"][ref["first"][i]:ref["first"][i] + ref["n_ent"][i]]
            assert np.array_equal(a, b), (i, ln, a[(a != b).any(axis=1)][:2], b[(a != b).any(axis=1)][:2])
    if min_handled is not None and n_sd:
        assert n_handled >= min_handled * n_sd, (n_handled, n_sd)
    return n_handled, n_sd


HDR = b"<13>1 2015-08-05T15:53:45.637824Z host app 1234 ID7 "


@pytest.mark.parametrize("geom", [dict(), dict(lines_per_group=33, tile_cap=16384), dict(lines_per_group=7, tile_cap=6144),
                                  dict(lines_per_group=64, tile_cap=36864), dict(head_cap=1024, tile_cap=18432)])
def test_corpus(lib, geom):
    """the BASELINE configs[3] corpus: (nearly) every line is handled -- the invalid ones with their exact error"""
    lines = synth.rfc5424_lines(6000, cfg=4, sd=True)
    h, n = check(lib, lines, min_handled=0.985, **geom)


def test_long_tail_heads(lib):
    lines = synth.rfc5424_lines(3000, cfg=5, sd=True, long_tail=True)
    check(lib, lines, head_cap=1024, tile_cap=18432)
    check(lib, lines, tile_cap=18432)


def test_shapes(lib):
    """hand-written shapes: everyday, tolerated oddities (handed back or decided exactly), every error of the structured data"""
    bodies = [
        b'[a b="c"] m', b'[a b="c"][d e="f"] m', b'[a b="c" d="e"] m', b'[a b="c"]', b'[a b="c"]x', b'[a b="c"][', b'[a b="c"][d', b'[a b="c"][d ] m',
        b'[a b="c"c="d"] m', b'[a "b="c"] m', b'[id ] m', b'[id] m', b'[a b= "c"] m', b'[a b="c" m', b'[a b="c', b'[a b="c\\"] m', b'[a b="c\\\\"] m',
        b'[a b="x\\\\" c="\\]\\q\\""] m', b'[a  b="c"] m', b'[a b="c"  d="e"] m', b'[a b="c" ] m', b'[a b="c"  ] m', b'[a b="c"] [d e="f"] m',
        b'[a b="c"]  m', b'[ b="c"] m', b'[a =="c"] m', b'[a b=="c"] m', b'[a b\\="c"] m', b'[a b="c"\\] m', b'[a\\ b="c"] m', b'[a\\" b="c"] m',
        b'[a b="c" \\"d="e"] m', b'[a b="c" \\\\"d="e"] m', b'[a b="' + b"v" * 70 + b'"] m', b'[a ' + b"n" * 70 + b'="v"] m', b'[' + b"i" * 70 + b' b="c"] m',
        b'[a b="c"]' + b" " * 70 + b'm', b'[a b="c"' + b" " * 70 + b'] m', "[a b=\"é\\\"]中\"][c d=\"e\"]  trailing  ".encode(), b'[a b="c"][d e="f"][g h="i"] m',
        b'[a ' + b" ".join(b'k%d="v%d"' % (k, k) for k in range(70)) + b'] m', b'[a ' + b"".join(b'k%d="v"' % k for k in range(5)) + b'] m',
        b'[a b="' + b"\\" * 15 + b'"] m', b'[a b="' + b"\\" * 16 + b'"] m', b'[a b="' + b"\\" * 17 + b'" c="d"] m', b'[a b="' + b"\\" * 33 + b'"] m',
        b'[a b="c"] "quoted" message "with" quotes', b'[a b="c"] one " quote', b'[a b="c" d="] m', b'[a b="c" d="e] f"] m', b'[a b="]"] m', b'[a b="[x]"] m',
        b'[nospace]', b'[id k= "v"] msg', b'[id k="v" msg', b'[a b="c"]]', b'[a b="c"] ]', b'[a]b="c"] m', b'[a b]="c"] m', b'[a b="c"\x7f] m', b'[a \tb="c"] m',
    ]
    lines = [HDR + b for b in bodies]
    # every shape at several positions of a tile (between everyday lines), and as a group of its own
    filler = synth.rfc5424_lines(64, cfg=4, sd=True, invalid_frac=0)
    mixed = []
    for k, ln in enumerate(lines):
        mixed += filler[(k * 3) % 60:(k * 3) % 60 + 1 + k % 3] + [ln]
    for geom in (dict(), dict(lines_per_group=1), dict(lines_per_group=5, tile_cap=4096), dict(head_cap=1024)):
        check(lib, lines, **geom)
        check(lib, mixed, **geom)
    # what MUST be handled by the fast form (the corpus' shapes), exactly
    got, ref = walk(lib, [HDR + b for b in (b'[a b="c"] m', b'[a b="c"][d e="f"] m', b'[a b="x\\\\" c="\\]\\q\\""] m', b'[id k= "v"] msg',
                                           b'[id k="v" msg', b'[nospace]', b'[a b="c"]', b'[a b="c"]x')])
    assert got["handled"].all()
    assert list(got["status"]) == [0, 0, 0, 16, 17, 15, 13, 14]


def test_every_prefix_of_a_line(lib):
    ln = synth.rfc5424_lines(40, cfg=4, sd=True, invalid_frac=0)[17]
    p = sd_pos_of(ln)
    lines = [ln[:k] for k in range(p + 1, len(ln) + 1)]
    check(lib, lines)
    check(lib, lines, lines_per_group=3, tile_cap=4096)


def test_mutations(lib):
    rng = np.random.default_rng(52)
    base = synth.rfc5424_lines(4000, cfg=4, sd=True, invalid_frac=0)
    alphabet = [b" ", b"[", b"]", b'"', b"\\", b"=", b"<", b">", b"-", b"a", b"", b"  ", b'""', b"\\\\", b'\\"', b"] ", b"][", b'="', b'" ', "é".encode(),
                b"\x7f", b"\t"]
    lines = []
    for ln in base:
        a, z = ln.find(b"["), ln.rfind(b"]")
        b = bytearray(ln)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(a, min(z + 8, len(b))))
            if pos < len(b) and b[pos] < 0x80:
                b[pos:pos + 1] = alphabet[int(rng.integers(0, len(alphabet)))]
        lines.append(bytes(b))
    h, n = check(lib, lines)
    assert h > 0.5 * n  # (most mutations are decided by the fast form itself: an exact error, or still a valid line)
    check(lib, lines[:1500], lines_per_group=9, tile_cap=8192)
    check(lib, lines[:1500], head_cap=1024)
