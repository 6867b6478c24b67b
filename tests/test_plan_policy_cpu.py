"""CPU: the launch policies of the streaming decoders (flowgger_amd/csrc/fg_plan_policy.hpp) -- how a batch is cut into chunks, when
chunks are drawn by ticket, how many entry slots a wave reserves.  Pure host arithmetic, swept here; what the kernels make of a plan
is the -m gpu suite's business (tests/test_gpu_round5.py: every format at 1 .. 70 000 lines under both dispatch forms)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

from flowgger_amd import _lib as L

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests/native/plan_host.cpp"
LIB = ROOT / "tests/native/libplan_host.so"
DEPS = [SRC, ROOT / "flowgger_amd/csrc/fg_plan_policy.hpp", ROOT / "include/fg_hip.h"]
u64, u32 = C.c_uint64, C.c_uint32


@pytest.fixture(scope="module")
def plan():
    if not LIB.exists() or LIB.stat().st_mtime < max(p.stat().st_mtime for p in DEPS):
        subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-Wall", "-o", str(LIB), str(SRC)], check=True)
    lib = C.CDLL(str(LIB))
    lib.fgp_plan_chunks.argtypes = [u64, u64, u32, u64, u64, u32, u32, u32, C.POINTER(u64)]
    lib.fgp_entry_chunk.argtypes = [u64, u32, u64, u32]
    lib.fgp_entry_chunk.restype = u32

    def chunks(n, blocks, L_=64, g=64, full=256, ticket_from=2, flags=0, chunk_lines=0):
        out = (u64 * 4)()
        lib.fgp_plan_chunks(n, blocks, L_, g, full, ticket_from, flags, chunk_lines, out)
        return int(out[0]), int(out[1]), int(out[2]), bool(out[3])

    return chunks, lib.fgp_entry_chunk


GEOMETRIES = [  # (L, g, full, ticket_from, blocks): headline, structured data, GELF, LTSV long lines
    (64, 64, 256, 20, 1792), (64, 20, 1024, 2, 2048), (8, 8, 128, 2, 5120), (64, 30, 512, 2, 1792), (64, 1, 512, 2, 1792)]


def test_every_plan_covers_the_batch_and_fits_the_grid(plan):
    chunks, _ = plan
    rng = np.random.default_rng(5)
    sizes = [1, 2, 7, 63, 64, 65, 1000, 16_384, 65_536, 262_144, 1 << 20, 4 << 20, 16 << 20, 100_000_000, 1 << 32] + \
        [int(x) for x in np.exp(rng.uniform(0, np.log(2e9), 300))]
    for L_, g, full, tf, blocks in GEOMETRIES:
        for flags in (0, L.FG_LO_STATIC_CHUNKS):
            for cl in (0, 1, 3, 64, 100, 65536):
                for n in sizes:
                    chunk, nch, blk, tickets = chunks(n, blocks, L_, g, full, tf, flags, cl)
                    assert chunk >= 1 and nch == -(-n // chunk), (n, L_, g, full, flags, cl)
                    assert 1 <= blk <= min(blocks, nch)
                    assert not (tickets and flags & L.FG_LO_STATIC_CHUNKS)
                    if not tickets and not flags and not cl:
                        assert nch <= blocks, "without tickets a wave takes ONE chunk: no more chunks than waves"
                    if flags & L.FG_LO_STATIC_CHUNKS and not (cl >= L_):
                        assert chunk >= min(L_, 65536), "the round-robin form never cuts below a wave's width"


def test_the_documented_cases(plan):
    chunks, _ = plan
    # headline kernel (64 lines to the group, 1792 waves, tickets from 20 chunks of 256 per wave)
    assert chunks(100_000_000, 1792, 64, 64, 256, 20) == (256, 390_625, 1792, True)
    assert chunks(16 << 20, 1792, 64, 64, 256, 20)[3] is True           # 9363 lines per wave >= 5120
    c = chunks(4 << 20, 1792, 64, 64, 256, 20)
    assert c == (2341, 1792, 1792, False)                                # one chunk per wave, no tickets
    assert chunks(65_536, 1024, 64, 64, 256, 20) == (64, 1024, 1024, False)
    assert chunks(1, 1792, 64, 64, 256, 20) == (64, 1, 1, False)
    # structured data, 21 lines to a 12 KiB tile: a 16 K-line batch is cut down to one (short) group per wave, not to 64 lines
    chunk, nch, blk, tickets = chunks(16_384, 820, 64, 20, 1024, 2)
    assert (chunk, tickets) == (20, False) and nch == blk == 820
    # ... and the round-robin form of rounds 3-4 kept its 64-line floor
    assert chunks(16_384, 820, 64, 20, 1024, 2, L.FG_LO_STATIC_CHUNKS)[0] == 64
    # GELF: tickets from two chunks of 128 lines per wave on
    assert chunks(4_000_000, 5120, 8, 8, 128, 2) == (128, 31_250, 5120, True)
    assert chunks(524_288, 5120, 8, 8, 128, 2)[3] is False
    # a named chunk size under dynamic dispatch always draws (tests, tuning)
    assert chunks(70_000, 1792, 64, 64, 256, 20, 0, 3) == (40, 1750, 1750, True) or chunks(70_000, 1792, 64, 64, 256, 20, 0, 3)[3] is True


def test_entry_reservations(plan):
    _, entry_chunk = plan
    # a large table: the share per wave, at most 4096, capped by eight slots per line of this launch, never below 256
    assert entry_chunk(400_000_000, 2048, 125_000_000, 0) == 4096
    assert entry_chunk(16_800_000, 5120, 65_536, 0) == 256      # a small GELF launch: 13 lines per wave -- still chunked (round 4: exact)
    assert entry_chunk(16_800_000, 5120, 4_000_000, 0) == 256   # 16.8 M / (16 * 5120) = 205 -> the floor
    assert entry_chunk(200_000, 2048, 60_000, 0) == 0           # a table sized tightly: exact reservations
    assert entry_chunk(0, 1792, 1000, 0) == 0
    assert entry_chunk(1 << 30, 7, 700, 0) == 800               # eight slots per line of the wave's share (100 lines)
    # the caller's word
    assert entry_chunk(1 << 30, 2048, 1 << 20, 1) == 0 and entry_chunk(1 << 30, 2048, 1 << 20, 777) == 777
    # never more than a quarter of the table stranded in the worst case
    rng = np.random.default_rng(7)
    for _ in range(2000):
        cap, waves, n = int(rng.integers(0, 1 << 31)), int(rng.integers(1, 8192)), int(rng.integers(1, 1 << 28))
        c = entry_chunk(cap, waves, n, 0)
        assert c == 0 or (256 <= c <= 4096 and c * waves <= cap // 4 + 4096 * waves // 16 + 256 * waves)
        if c:
            assert cap // (4 * waves) >= 256
