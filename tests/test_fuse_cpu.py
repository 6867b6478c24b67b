"""Framing inside the decode kernels (flowgger_amd/csrc/fg_fuse.hpp) on the CPU emulation of a wave against the oracle's restatement of
LineSplitter / NulSplitter (fgo_frame; src/flowgger/splitter/line_splitter.rs:17-25, nul_splitter.rs:18-40): every frame's start, end
and UTF-8 verdict, for tiles from 16 bytes up, look-ahead from 0 up, streams that end with and without a terminator, final and
non-final chunks, lines longer than the tile, more lines in a tile than a pass takes, and damage at every kind of boundary."""
from __future__ import annotations

import random

import numpy as np
import pytest

from fuse_binding import FuseHost
from oracle_binding import Oracle


@pytest.fixture(scope="module")
def host():
    return FuseHost()


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


def expect(oracle, raw: bytes, framing: str, final: bool):
    fr = oracle.frame(raw, framing)
    if not final and fr and fr[-1][1] == len(raw) and not raw.endswith(b"\n" if framing == "line" else b"\0"):
        fr = fr[:-1]  # an unterminated piece stays with the caller
    starts = np.array([f[0] for f in fr], np.uint64)
    ends = np.array([f[1] for f in fr], np.uint64)
    bad = np.array([0 if f[3] else 1 for f in fr], np.uint8)
    return starts, ends, bad


def check(host, oracle, raw: bytes, framing: str, final: bool, S: int, look: int, tile_cap: int, lines: int = 64):
    delim = 0x0A if framing == "line" else 0
    for garbage in (delim, 0xC3, 0x80):
        gs, ge, gb, consumed, passes, scans = host.frame(raw, delim, final, S, look, tile_cap, lines, garbage)
        ws, we, wb = expect(oracle, raw, framing, final)
        ctx = f"S={S} look={look} lines={lines} final={final} framing={framing} len={len(raw)}"
        assert len(gs) == len(ws), f"{ctx}: {len(gs)} frames, the oracle has {len(ws)}"
        assert np.array_equal(gs, ws) and np.array_equal(ge, we), ctx
        assert np.array_equal(gb, wb), f"{ctx}: UTF-8 verdicts differ at frame {int(np.flatnonzero(gb != wb)[0])}"
        assert consumed == (int(we[-1]) if len(we) else 0), ctx
    return passes, scans


HAND = [
    b"", b"\n", b"a", b"a\n", b"\n\n\n", b"a\nb", b"a\r\nb\r\n", b"\r\n", b"abc\n" * 40, b"x" * 100, b"x" * 100 + b"\n", b"\n" + b"y" * 333,
    "é\n".encode(), "é".encode(), b"\xc3", b"\xc3\n", b"a\xc3", b"ab\xe2\x82", b"ab\xe2\x82\n\xac\n", b"\xf0\x9f\x98\x80\n", b"\xf0\x9f\x98",
    b"\x80\n", b"\n\x80", b"\xed\xa0\x80\n", b"\xc0\xaf\n", b"\xf5\n", b"ok\n\xff\nok\n",
]


def test_hand_written_streams_every_small_geometry(host, oracle):
    for raw in HAND:
        for framing in ("line", "nul"):
            data = raw if framing == "line" else raw.replace(b"\n", b"\0")
            for final in (True, False):
                for S, look in ((16, 0), (16, 16), (32, 16), (48, 64), (256, 64)):
                    check(host, oracle, data, framing, final, S, look, 1024)


def soup(rng: random.Random, n: int, mean: int, damage: float) -> bytes:
    out = bytearray()
    while len(out) < n:
        kind = rng.random()
        ln = 0 if kind < 0.05 else int(rng.expovariate(1.0 / mean)) if kind < 0.9 else rng.randrange(mean * 8, mean * 40)
        body = bytearray(rng.choice(b"abcdefghijklmnop qrstuvwxyz0123456789<>[]=\"") for _ in range(ln))
        for _ in range(ln // 40):
            at = rng.randrange(0, max(ln, 1))
            body[at:at] = rng.choice(["é", "€", "😀", "　"]).encode()
        if rng.random() < damage and len(body):
            at = rng.randrange(0, len(body))
            body[at:at + 1] = bytes([rng.choice([0x80, 0xC3, 0xE2, 0xF0, 0xFF, 0xC0, 0xED])])
        if rng.random() < 0.1:
            body += b"\r"
        out += body + b"\n"
    return bytes(out)


@pytest.mark.parametrize("seed", range(6))
def test_random_streams_against_the_oracle(host, oracle, seed):
    rng = random.Random(0xF05E + seed)
    raw = soup(rng, 6000 + 3000 * seed, (12, 40, 90, 254, 300, 600)[seed], 0.08)
    for cut in (len(raw), len(raw) - rng.randrange(1, 50), len(raw) - rng.randrange(50, 400)):
        data = raw[:cut]
        for framing in ("line", "nul"):
            d = data if framing == "line" else data.replace(b"\n", b"\0")
            for final in (True, False):
                for S, look, lines in ((16, 0, 64), (64, 16, 8), (256, 64, 64), (1024, 256, 64), (2288, 320, 8), (4096 - 96, 64, 64)):
                    check(host, oracle, d, framing, final, S, look, 4096, lines)


def test_many_short_lines_take_several_lists_and_passes(host, oracle):
    raw = b"\n".join(b"l%d" % (i % 7) for i in range(3000)) + b"\n"
    passes, scans = check(host, oracle, raw, "line", True, 3840, 64, 4096)
    assert passes > len(raw) // 3840  # tiles of ~1100 lines: 256 to the list, 64 to the pass
    check(host, oracle, raw, "line", True, 3840, 64, 4096, lines=8)


def test_lines_longer_than_the_look_ahead_are_finished_by_the_forward_scan(host, oracle):
    rng = random.Random(7)
    raw = soup(rng, 40000, 700, 0.05)
    passes, scans = check(host, oracle, raw, "line", True, 1024, 64, 2048)
    assert scans > 5
    check(host, oracle, raw[:-1], "line", False, 1024, 64, 2048)
    # one line that spans many tiles and the end of the stream, damaged just before the end
    long = b"head\n" + b"z" * 9000 + b"\xe2\x82"
    check(host, oracle, long, "line", True, 512, 64, 1024)
    check(host, oracle, long, "line", False, 512, 64, 1024)


def test_the_planned_geometries(host, oracle):
    rng = random.Random(11)
    for avg, lines, cap in ((254, 64, 20480), (307, 8, 3072), (554, 64, 18432), (64, 64, 8192), (3000, 64, 57344)):
        S, look = host.plan(avg, lines, cap)
        assert S % 16 == 0 and look % 16 == 0 and 16 + S + look + 16 <= cap and S >= 256
        raw = soup(rng, 5 * cap, avg, 0.02)
        check(host, oracle, raw, "line", True, S, look, cap, lines)
