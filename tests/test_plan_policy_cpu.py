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
lib_holder = []


@pytest.fixture(scope="module")
def plan():
    if not LIB.exists() or LIB.stat().st_mtime < max(p.stat().st_mtime for p in DEPS):
        subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-Wall", "-o", str(LIB), str(SRC)], check=True)
    lib = C.CDLL(str(LIB))
    lib.fgp_plan_chunks.argtypes = [u64, u64, u32, u64, u64, u32, u32, u32, u32, C.POINTER(u64)]
    lib.fgp_entry_chunk.argtypes = [u64, u32, u64, u32]
    lib.fgp_entry_chunk.restype = u32
    lib.fgp_entry_chunk_shared.argtypes = [u64, u32, u64, u32, u32]
    lib.fgp_entry_chunk_shared.restype = u32
    lib_holder.append(lib)
    lib.fgp_chunk_range.argtypes = [u64, u64, u32, u32, u32, u64, C.POINTER(u64)]

    def chunks(n, blocks, L_=64, g=64, full=256, ticket_from=2, flags=0, chunk_lines=0, taper=False, levels=1):
        out = (u64 * 7)()
        lib.fgp_plan_chunks(n, blocks, L_, g, full, ticket_from, flags, chunk_lines, levels, out)
        r = int(out[0]), int(out[1]), int(out[2]), bool(out[3])
        return r + (tuple(int(out[4 + j]) for j in range(3)),) if taper else r

    def chunk_range(c, chunk, taper, n):
        out = (u64 * 2)()
        lib.fgp_chunk_range(c, chunk, taper[0], taper[1], taper[2], n, out)
        return int(out[0]), int(out[1])

    chunks.range = chunk_range
    return chunks, lib.fgp_entry_chunk


NO_TAPER = 0xFFFFFFFF
GEOMETRIES = [  # (L, g, full, ticket_from, blocks, taper levels): headline, structured data, GELF, LTSV long lines
    (64, 64, 256, 20, 1792, 0), (64, 20, 128, 2, 2048, 1), (8, 8, 64, 2, 5120, 1), (64, 30, 512, 2, 1792, 1), (64, 1, 512, 2, 1792, 3),
    (64, 20, 1024, 2, 2048, 3), (8, 8, 128, 2, 5120, 2)]


def test_every_plan_covers_the_batch_and_fits_the_grid(plan):
    chunks, _ = plan
    rng = np.random.default_rng(5)
    sizes = [1, 2, 7, 63, 64, 65, 1000, 16_384, 65_536, 262_144, 1 << 20, 4 << 20, 16 << 20, 100_000_000, 1 << 32] + \
        [int(x) for x in np.exp(rng.uniform(0, np.log(2e9), 300))]
    for L_, g, full, tf, blocks, lv in GEOMETRIES:
        for flags in (0, L.FG_LO_STATIC_CHUNKS, L.FG_LO_TAPER_1 | L.FG_LO_TAPER_2):
            for cl in (0, 1, 3, 64, 100, 65536):
                for n in sizes:
                    chunk, nch, blk, tickets, taper = chunks(n, blocks, L_, g, full, tf, flags, cl, taper=True, levels=lv)
                    assert chunk >= 1 and nch >= -(-n // chunk), (n, L_, g, full, flags, cl)
                    if lv == 0 and not flags & (L.FG_LO_TAPER_1 | L.FG_LO_TAPER_2):
                        assert taper[0] == NO_TAPER, "the headline kernel never tapers by itself"
                    if taper[0] == NO_TAPER:
                        assert nch == -(-n // chunk) and taper == (NO_TAPER,) * 3
                    else:  # the tail is at most one chunk per wave's worth of lines, in at most 4 x as many chunks
                        assert tickets and nch <= -(-n // chunk) + 7 * min(blocks, n // (2 * chunk)) + 8
                        assert chunks(n, blocks, L_, g, full, tf, flags | L.FG_LO_NO_TAPER, cl, levels=lv) == (chunk, -(-n // chunk), min(blocks, -(-n // chunk)), True)
                    assert 1 <= blk <= min(blocks, nch)
                    assert not (tickets and flags & L.FG_LO_STATIC_CHUNKS)
                    if not tickets and not flags and not cl:
                        assert nch <= blocks, "without tickets a wave takes ONE chunk: no more chunks than waves"
                    if flags & L.FG_LO_STATIC_CHUNKS and not (cl >= L_):
                        assert chunk >= min(L_, 65536), "the round-robin form never cuts below a wave's width"


def test_the_documented_cases(plan):
    chunks, _ = plan
    # headline kernel (64 lines to the group, 1792 waves, tickets from 20 chunks of 256 per wave)
    assert chunks(100_000_000, 1792, 64, 64, 256, 20, levels=0) == (256, 390_625, 1792, True)
    # ... under FG_LO_TAPER_1 | _2 (tuning): the last chunk per wave's worth of lines in halves and quarters (a chunk never falls below
    # one average group: 64 lines)
    assert chunks(100_000_000, 1792, 64, 64, 256, 20, L.FG_LO_TAPER_1 | L.FG_LO_TAPER_2, taper=True, levels=0) == \
        (256, 388_833 + 1792 + 2 * 1792, 1792, True, (388_833, 390_625, NO_TAPER))
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
    # GELF: tickets from two chunks of 64 lines per wave on; the last 5120 chunks' worth of lines go out as 10 240 chunks of 32
    assert chunks(4_000_000, 5120, 8, 8, 64, 2, L.FG_LO_NO_TAPER) == (64, 62_500, 5120, True)
    assert chunks(4_000_000, 5120, 8, 8, 64, 2, taper=True) == (64, 57_380 + 10_240, 5120, True, (57_380, NO_TAPER, NO_TAPER))
    # ... three levels (tuning): 5120 chunks of 64, 5120 of 32 and 10 240 of 16 lines behind 26 130 of 128
    assert chunks(4_000_000, 5120, 8, 8, 128, 2, taper=True, levels=3) == (128, 26_130 + 5120 + 5120 + 10_240, 5120, True, (26_130, 31_250, 36_370))
    assert chunks(524_288, 5120, 8, 8, 64, 2)[3] is False
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


def test_tapered_chunks_tile_the_batch_exactly(plan):
    """chunk index -> lines (fg::chunk_range, the function the kernels call): consecutive, non-empty, in order, covering [0, n) with
    exactly the number of chunks the host books on the ticket counter."""
    chunks, _ = plan
    rng = np.random.default_rng(11)
    cases = [(4_000_000, 5120, 8, 8, 128, 2, 0), (70_000, 5120, 8, 8, 128, 2, 64), (70_000, 5120, 8, 8, 128, 2, 16), (4_194_304, 2048, 64, 20, 1024, 2, 512),
             (1_000_003, 1792, 64, 64, 256, 2, 0), (999, 1792, 64, 1, 512, 2, 8), (100_000, 7, 64, 5, 512, 2, 0)]
    for _ in range(60):
        L_ = int(rng.choice([8, 64]))
        cases.append((int(rng.integers(1, 3_000_000)), int(rng.integers(1, 6000)), L_, int(rng.integers(1, L_ + 1)),
                      int(rng.choice([128, 256, 512, 1024])), 2, int(rng.choice([0, 0, 5, 64, 100, 777]))))
    tapered = 0
    for idx, (n, blocks, L_, g, full, tf, cl) in enumerate(cases):
        chunk, nch, blk, tickets, taper = chunks(n, blocks, L_, g, full, tf, 0, cl, taper=True, levels=1 + idx % 3)
        tapered += taper[0] != NO_TAPER
        if nch > 400_000:
            continue
        at = 0
        sizes = []
        for c in range(nch):
            lo, hi = chunks.range(c, chunk, taper, n)
            assert lo == at and hi > lo, (n, blocks, chunk, taper, c)
            sizes.append(hi - lo)
            at = hi
        assert at == n
        assert chunks.range(nch, chunk, taper, n) == (n, n) and chunks.range(nch + 12345, chunk, taper, n) == (n, n)
        assert all(a >= b for a, b in zip(sizes[:-1], sizes[1:-1])), "chunk sizes never grow (but for the batch's last, ragged one)"
        if taper[0] != NO_TAPER:
            assert min(sizes[:-1] or [g]) >= min(g, chunk), "no chunk below one average group"
    assert tapered >= 20


def test_slices_that_share_a_table_never_strand_more_than_a_quarter_of_it(plan):
    """the sliced host paths decode ONE batch as 6 .. 64 launches into ONE entry table (nbytes / 16 or / 8 slots); every launch
    strands at most waves x chunk slots.  With the table's budget divided by the launches that share it, what ALL slices strand stays
    below a quarter of the table -- the floor of 256 slots per reservation included -- for every batch size and grid the paths use."""
    shared = lib_holder[0].fgp_entry_chunk_shared
    for nbytes in (48 << 20, 256 << 20, 1 << 30, 4 << 30):
        for per_byte in (16, 8):
            ent_cap = nbytes // per_byte + 1024
            for slices in (6, 16, 32, 64):
                for blocks in (1792, 2048, 4096, 5120):
                    lines = nbytes // 300 // slices
                    c = shared(ent_cap, blocks, lines, 0, slices)
                    stranded = slices * min(blocks, max(lines, 1)) * c  # every wave of every slice, a whole reservation
                    assert stranded <= ent_cap // 4 + blocks, (nbytes, per_byte, slices, blocks, c)
                    assert c == 0 or c >= 256
    # ... and a launch alone keeps the round-5 rule
    assert shared(400_000_000, 2048, 125_000_000, 0, 1) == 4096
