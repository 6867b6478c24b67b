"""CPU: the HOST side of the C ABI -- flowgger_amd/csrc/fg_capi.cpp + fg_host_pipeline.cpp themselves, compiled by g++ against a synchronous stand-in for the
HIP runtime (tests/native/fakehip) and fake kernel launchers with the real kernels' contract (tests/native/host_pipeline_fake.cpp).
What is tested is the bookkeeping of the host paths: slices cut at line boundaries, rows at their final index, the entry columns
brought back per slice as ranges of one shared counter, the retry when the entry table is too small, the raw-stream path's
per-slice frame counts and its fall-back to the one-piece form, unterminated tails, error paths.  The sliced paths must give what
ONE call of the device entry point gives on the same batch.  (What the GPU computes is the -m gpu suite's business.)"""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

from flowgger_amd import _lib as L

ROOT = Path(__file__).resolve().parent.parent
HERE = ROOT / "tests" / "native"
LIB = HERE / "libhost_pipeline_fake.so"
SRC = [HERE / "host_pipeline_fake.cpp", HERE / "fakehip/hip/hip_runtime.h"] + [ROOT / "flowgger_amd/csrc" / f for f in
                                                                                ("fg_capi.cpp", "fg_host_pipeline.cpp", "fg_ctx.hpp", "fg_fused_plan.hpp", "fg_tile_cap.hpp", "fg_gather.cpp", "fg_materialize.cpp")] + [ROOT / "include/fg_hip.h"]
u64, vp = C.c_uint64, C.c_void_p
COLS = ["meta", "ts", "hostname", "appname", "procid", "msgid", "msg", "full_msg", "ent_count"]


@pytest.fixture(scope="module")
def fake():
    if not LIB.exists() or any(s.stat().st_mtime > LIB.stat().st_mtime for s in SRC):
        subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas",
                        "-Wno-unused-variable", f"-I{HERE / 'fakehip'}", f"-I{ROOT / 'include'}", "-o", str(LIB), str(HERE / "host_pipeline_fake.cpp")],
                       check=True)
    lib = C.CDLL(str(LIB))
    lib.fg_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    lib.fg_destroy.argtypes = [vp]
    lib.fg_destroy.restype = None
    lib.fg_set_launch_opts.argtypes = [vp, C.POINTER(L.fg_launch_opts)]
    lib.fg_tables_layout.argtypes = [u64, u64, C.POINTER(u64)]
    lib.fg_decode_batch_device.argtypes = [vp, C.c_int, vp, u64, vp, u64, C.POINTER(L.fg_tables), vp]
    lib.fg_decode_batch.argtypes = [vp, C.c_int, vp, u64, vp, u64, C.POINTER(L.fg_tables)]
    lib.fg_frame_decode_batch.argtypes = [vp, C.c_int, C.c_int, vp, u64, C.c_int, C.POINTER(L.fg_tables), C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    lib.fgf_launches.restype = C.c_ulonglong
    lib.fgf_fail_malloc_after.argtypes = [C.c_longlong]
    return lib


class Ctx:
    def __init__(self, lib):
        self.lib, self.h = lib, vp()
        assert lib.fg_create(0, None, C.byref(self.h)) == 0

    def close(self):
        self.lib.fg_destroy(self.h)


def snapshot(st, n):
    """per-line content of ctx-owned tables: fixed columns + the line's entries"""
    used = int(np.ctypeslib.as_array(C.cast(st.ent_used, C.POINTER(C.c_uint64)), (1,))[0])
    def arr(name, dt, cnt):
        p = getattr(st, name)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (cnt * np.dtype(dt).itemsize,)).view(dt).copy() if cnt else np.zeros(0, dt)
    out = {"meta": arr("meta", np.uint32, n), "ts": arr("ts", np.float64, n), "ent_first": arr("ent_first", np.uint32, n),
           "ent_count": arr("ent_count", np.uint32, n)}
    for c in ("hostname", "appname", "procid", "msgid", "msg", "full_msg"):
        out[c] = arr(c, np.uint64, n)
    out["ent_name"], out["ent_val"] = arr("ent_name", np.uint64, used), arr("ent_val", np.uint64, used)
    out["ent_type"], out["ent_flags"] = arr("ent_type", np.uint8, used), arr("ent_flags", np.uint8, used)
    out["used"] = used
    return out


def device_reference(lib, data, offsets, ent_cap):
    """one call of the device entry point on the whole batch (host arrays play the device)"""
    n = len(offsets) - 1
    sizes = (u64 * L.FG_TABLE_ARRAYS)()
    assert lib.fg_tables_layout(n, ent_cap, sizes) == 0
    bufs = [np.zeros(max(int(s), 8), np.uint8) for s in sizes]
    st = L.fg_tables()
    st.n, st.ent_cap = n, ent_cap
    for name, b in zip(L.TABLE_FIELDS, bufs):
        setattr(st, name, b.ctypes.data)
    c = Ctx(lib)
    pad = np.concatenate([data, np.zeros(64, np.uint8)])
    assert lib.fg_decode_batch_device(c.h, 0, pad.ctypes.data, data.size, offsets.ctypes.data, n, C.byref(st), None) == 0
    snap = snapshot(st, n)
    c.close()
    snap["_keep"] = bufs
    return snap


def same_lines(a, b, n):
    for c in COLS:
        assert np.array_equal(a[c][:n], b[c][:n]), c
    # the entries of every line, wherever its slice of the table lies
    for i in np.nonzero(a["ent_count"][:n])[0]:
        fa, fb, k = int(a["ent_first"][i]), int(b["ent_first"][i]), int(a["ent_count"][i])
        for c in ("ent_name", "ent_val", "ent_type", "ent_flags"):
            assert np.array_equal(a[c][fa:fa + k], b[c][fb:fb + k]), (c, int(i))


def corpus(n, rng, eq_lo=0, eq_hi=6, length=240):
    lines = []
    for i in range(n):
        k = int(rng.integers(eq_lo, eq_hi + 1))
        body = (b"<%d>" % (i % 190)) + b" ".join(b'k%d="v%d"' % (j, (i + j) % 97) for j in range(k))
        lines.append(body + b" " + b"m" * max(length - len(body), 1) if i % 1009 else b"")
    return lines


def pack(lines):
    offsets = np.zeros(len(lines) + 1, np.uint64)
    np.cumsum([len(x) for x in lines], out=offsets[1:])
    return np.frombuffer(b"".join(lines), np.uint8).copy(), offsets


def test_sliced_decode_batch_equals_one_launch(fake):
    rng = np.random.default_rng(1)
    lines = corpus(330_000, rng)
    data, offsets = pack(lines)
    assert data.size > 2 * (32 << 20)
    n = len(lines)
    c = Ctx(fake)
    st = L.fg_tables()
    cnt = (C.c_ulonglong * 3)()
    fake.fgf_launches(1)
    fake.fgf_counters(cnt, 1)
    assert fake.fg_decode_batch(c.h, 0, data.ctypes.data, data.size, offsets.ctypes.data, n, C.byref(st)) == 0
    launches = fake.fgf_launches(1)
    fake.fgf_counters(cnt, 1)
    got = snapshot(st, n)
    want = device_reference(fake, data, offsets, data.size // 16 + 1024)
    assert launches >= 3 and got["used"] == want["used"] == int(got["ent_count"].sum())
    same_lines(got, want, n)
    # every table byte crossed the link once: rows (76 B) + entries (18 B) + the counters, nothing twice
    assert cnt[2] <= n * 76 + got["used"] * 18 + 64 * launches, (int(cnt[2]), n * 76 + got["used"] * 18)
    assert cnt[1] <= data.size + 16 * launches + (n + 1) * 8 + 64
    # a second, smaller batch on the same ctx (buffers are reused; one slice: the single-stream path)
    small, soffs = pack(lines[:5000])
    assert fake.fg_decode_batch(c.h, 0, small.ctypes.data, small.size, soffs.ctypes.data, 5000, C.byref(st)) == 0
    same_lines(snapshot(st, 5000), device_reference(fake, small, soffs, small.size // 16 + 1024), 5000)
    c.close()


def test_pinned_buffers_take_the_zero_copy_form(fake):
    """bytes + offsets in pinned memory: ONE launch reads them in place and writes the pinned tables -- no table byte and no line byte
    crosses the link by hipMemcpy (only the 8-byte entry counter comes back); the result is what the device entry point gives.  A too
    small entry table is retried with what the counter asked for; FG_LO_NO_ZERO_COPY and pageable buffers take the sliced pipeline."""
    rng = np.random.default_rng(4)
    lines = corpus(150_000, rng) + corpus(30_000, rng, 30, 40, 300)  # (the tail needs more entries than one per 16 bytes)
    data, offsets = pack(lines)
    n = len(lines)
    fake.fg_alloc_pinned.argtypes = [u64, C.POINTER(vp)]
    fake.fg_free_pinned.argtypes = [vp]
    pb, po = vp(), vp()
    assert fake.fg_alloc_pinned(data.size + 64, C.byref(pb)) == 0 and fake.fg_alloc_pinned((n + 1) * 8, C.byref(po)) == 0
    hb = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_uint8)), (data.size + 64,))
    ho = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), (n + 1,))
    hb[:data.size] = data
    ho[:] = offsets
    want = device_reference(fake, data, offsets, data.size // 4 + 1024)
    c = Ctx(fake)
    st = L.fg_tables()
    cnt = (C.c_ulonglong * 3)()
    fake.fgf_launches(1)
    fake.fgf_counters(cnt, 1)
    assert fake.fg_decode_batch(c.h, 0, pb, data.size, po, n, C.byref(st)) == 0
    launches = fake.fgf_launches(1)
    fake.fgf_counters(cnt, 1)
    got = snapshot(st, n)
    assert got["used"] == want["used"] and not (got["meta"] & 0xFF == 0xFE).any()
    same_lines(got, want, n)
    assert launches <= 2 and cnt[1] <= 64 and cnt[2] <= 64, (launches, int(cnt[1]), int(cnt[2]))  # (two launches: the entry-table retry)
    # the same pinned buffers with the flag: the sliced pipeline (bytes cross by hipMemcpy), same result
    lo = L.fg_launch_opts(0, 0, 0, 0, 0, L.FG_LO_NO_ZERO_COPY, 0)
    assert fake.fg_set_launch_opts(c.h, C.byref(lo)) == 0
    fake.fgf_counters(cnt, 1)
    assert fake.fg_decode_batch(c.h, 0, pb, data.size, po, n, C.byref(st)) == 0
    fake.fgf_counters(cnt, 1)
    assert cnt[1] >= data.size
    same_lines(snapshot(st, n), want, n)
    # pageable bytes, pinned offsets: not zero-copy
    assert fake.fg_set_launch_opts(c.h, None) == 0
    fake.fgf_counters(cnt, 1)
    assert fake.fg_decode_batch(c.h, 0, data.ctypes.data, data.size, po, n, C.byref(st)) == 0
    fake.fgf_counters(cnt, 1)
    assert cnt[1] >= data.size
    same_lines(snapshot(st, n), want, n)
    c.close()
    fake.fg_free_pinned(pb)
    fake.fg_free_pinned(po)


def test_a_few_short_lines_described_inside_a_large_buffer(fake):
    """fg_shard_plan gives EMPTY leading slices when the lines cover fewer bytes than there are slices (a few short
    lines -- or only empty ones -- whose offsets live in a >= 32 MiB buffer): the first slice WITH rows must upload its own start
    offset, or line 0 is decoded from a stale device value."""
    big = np.zeros(40 << 20, np.uint8)
    lines = [b'<1>a="b" c="d" x', b"", b'<2>e="f" y']
    blob = b"".join(lines)
    start = 1 << 20  # (not at the front of the buffer: a stale zero would be wrong)
    big[start:start + len(blob)] = np.frombuffer(blob, np.uint8)
    offsets = np.array([start, start + len(lines[0]), start + len(lines[0]), start + len(blob)], np.uint64)
    for offs in (offsets, np.full(6, start + 3, np.uint64)):  # (and: nothing but empty lines)
        n = len(offs) - 1
        c = Ctx(fake)
        st = L.fg_tables()
        # poison the ctx's device offsets with an earlier batch that started elsewhere
        warm, woffs = pack([b'<9>q="r" zzzz'] * 8)
        wbig = np.concatenate([np.zeros(64, np.uint8), warm, np.zeros(40 << 20, np.uint8)])
        assert fake.fg_decode_batch(c.h, 0, wbig.ctypes.data, wbig.size, (woffs + np.uint64(64)).ctypes.data, 8, C.byref(st)) == 0
        assert fake.fg_decode_batch(c.h, 0, big.ctypes.data, big.size, offs.ctypes.data, n, C.byref(st)) == 0
        same_lines(snapshot(st, n), device_reference(fake, big, offs, big.size // 16 + 1024), n)
        c.close()


def test_entry_table_too_small_is_noticed_mid_batch_and_the_retry_is_exact(fake):
    rng = np.random.default_rng(2)
    lines = corpus(60_000, rng, 0, 2) + corpus(150_000, rng, 30, 40, 300) + corpus(60_000, rng, 0, 2)  # > one entry per 16 bytes in the middle
    data, offsets = pack(lines)
    n = len(lines)
    assert data.size > (64 << 20)
    total = sum(ln.count(b"=") for ln in lines)
    assert total > data.size // 16 + 1024
    c = Ctx(fake)
    st = L.fg_tables()
    fake.fgf_launches(1)
    assert fake.fg_decode_batch(c.h, 0, data.ctypes.data, data.size, offsets.ctypes.data, n, C.byref(st)) == 0
    launches = fake.fgf_launches(1)
    got = snapshot(st, n)
    assert got["used"] == total and not (got["meta"] & 0xFF == 0xFE).any()
    sl = min(max(data.size // 8, 8 << 20), 32 << 20)  # (fg_host_pipeline.cpp's slice_count)
    slices = (data.size + sl - 1) // sl
    assert launches == 2 * slices  # every slice ran twice: once counting past the capacity, once with the capacity the counter asked for
    same_lines(got, device_reference(fake, data, offsets, total + 16), n)
    c.close()


@pytest.mark.parametrize("final", [1, 0])
@pytest.mark.parametrize("dense", [False, True])
def test_raw_stream_sliced_equals_one_piece(fake, final, dense):
    rng = np.random.default_rng(3)
    lines = [ln for ln in corpus(250_000, rng, 30 if dense else 0, 40 if dense else 5, 300 if dense else 240)]
    parts = [ln + (b"\r\n" if i % 7 == 0 else b"\n") for i, ln in enumerate(lines)]
    parts[1234] = b"\xff" + parts[1234]
    tail = b"<1>tail without a terminator a=1"
    raw = np.frombuffer(b"".join(parts) + tail, np.uint8).copy()
    assert raw.size > (48 << 20)
    res = {}
    for one_piece in (False, True):
        c = Ctx(fake)
        lo = L.fg_launch_opts()
        lo.flags = L.FG_LO_TRANSCODE_ONE_PIECE if one_piece else 0
        assert fake.fg_set_launch_opts(c.h, C.byref(lo)) == 0
        st, po, nf, cons = L.fg_tables(), vp(), u64(), u64()
        pad = np.concatenate([raw, np.zeros(64, np.uint8)])
        rc = fake.fg_frame_decode_batch(c.h, 0, 1, pad.ctypes.data, raw.size, final, C.byref(st), C.byref(po), C.byref(nf), C.byref(cons))
        assert rc == 0
        n = int(nf.value)
        offs = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), (n + 1,)).copy()
        res[one_piece] = (snapshot(st, n), offs, int(cons.value), n)
        c.close()
    (a, ao, ac, an), (b, bo, bc, bn) = res[False], res[True]
    assert an == bn == len(lines) + (1 if final else 0) and ac == bc == (raw.size if final else raw.size - len(tail))
    assert np.array_equal(ao, bo)
    same_lines(a, b, an)
    assert (a["meta"][1234] & 0xFF) == 0xFD and int((a["meta"] & 0xFF == 0xFD).sum()) == 1
    assert a["used"] == int(a["ent_count"].sum()) == b["used"]


@pytest.mark.parametrize("one_piece", [False, True])
def test_a_one_pass_framing_scan_that_gives_up_is_redone_by_the_three_kernel_form(fake, one_piece):
    """k_frame_onepass's look-back is bounded: a wave that waits too long on a predecessor's descriptor raises the abort word and the total
    comes back as FG_FRAME_ABORTED.  The sliced path then leaves for the one-piece path, which launches the framing again with
    classic = 1; the caller sees the same frames as ever."""
    rng = np.random.default_rng(5)
    lines = corpus(250_000, rng)
    raw = np.frombuffer(b"".join(ln + b"\n" for ln in lines) + b"<1>tail a=1", np.uint8).copy()
    assert raw.size > (48 << 20)
    pad = np.concatenate([raw, np.zeros(64, np.uint8)])
    fake.fake_frame_abort_next.argtypes = [C.c_int]
    res = {}
    for abort in (0, 3):
        c = Ctx(fake)
        lo = L.fg_launch_opts()
        lo.flags = L.FG_LO_TRANSCODE_ONE_PIECE if one_piece else 0
        assert fake.fg_set_launch_opts(c.h, C.byref(lo)) == 0
        fake.fake_frame_abort_next(abort)
        st, po, nf, cons = L.fg_tables(), vp(), u64(), u64()
        assert fake.fg_frame_decode_batch(c.h, 0, 1, pad.ctypes.data, raw.size, 1, C.byref(st), C.byref(po), C.byref(nf), C.byref(cons)) == 0
        n = int(nf.value)
        offs = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), (n + 1,)).copy()
        res[abort] = (snapshot(st, n), offs, int(cons.value), n)
        assert fake.fake_frame_classic_launches() == (1 if abort else 0)
        c.close()
    fake.fake_frame_abort_next(0)
    (a, ao, ac, an), (b, bo, bc, bn) = res[0], res[3]
    assert an == bn == len(lines) + 1 and ac == bc == raw.size and np.array_equal(ao, bo)
    same_lines(a, b, an)


def test_a_pinned_raw_stream_is_uploaded_by_the_framing_scan_itself(fake):
    """raw chunk in pinned memory: no hipMemcpy of the stream at all -- the framing scan reads it in place and stores it to its place on
    the device (k_frame_scan<COPY>), the tables are written straight into the pinned host tables; only frame offsets and a few
    counters are copied.  Same rows, offsets and entries as the pageable chunk through the hipMemcpy uploads."""
    rng = np.random.default_rng(8)
    lines = corpus(230_000, rng)
    raw = np.frombuffer(b"".join(ln + b"\n" for ln in lines) + b"<1>unterminated tail a=1", np.uint8).copy()
    assert raw.size > (48 << 20)
    fake.fg_alloc_pinned.argtypes = [u64, C.POINTER(vp)]
    fake.fg_free_pinned.argtypes = [vp]
    fake.fgf_kernel_uploaded.restype = C.c_ulonglong
    pb = vp()
    assert fake.fg_alloc_pinned(raw.size + 64, C.byref(pb)) == 0
    hb = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_uint8)), (raw.size + 64,))
    hb[:raw.size] = raw
    res = {}
    cnt = (C.c_ulonglong * 3)()
    for pinned in (True, False):
        c = Ctx(fake)
        lo = L.fg_launch_opts(0, 0, 0, 0, 0, L.FG_LO_FRAME_KERNEL_UPLOAD, 0)  # (off by default: measured slower than the copy engine)
        assert fake.fg_set_launch_opts(c.h, C.byref(lo)) == 0
        st, po, nf, cons = L.fg_tables(), vp(), u64(), u64()
        pad = np.concatenate([raw, np.zeros(64, np.uint8)])
        fake.fgf_kernel_uploaded(1)
        fake.fgf_counters(cnt, 1)
        src = pb if pinned else pad.ctypes.data
        assert fake.fg_frame_decode_batch(c.h, 0, 1, src, raw.size, 1, C.byref(st), C.byref(po), C.byref(nf), C.byref(cons)) == 0
        fake.fgf_counters(cnt, 1)
        n = int(nf.value)
        offs = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), (n + 1,)).copy()
        res[pinned] = (snapshot(st, n), offs, int(cons.value), n)
        up_by_kernel = fake.fgf_kernel_uploaded(1)
        if pinned:
            assert up_by_kernel >= raw.size and cnt[1] < (1 << 16), (up_by_kernel, int(cnt[1]))   # nothing of the stream by hipMemcpy
            assert cnt[2] <= (n + 1) * 8 + 4096, int(cnt[2])                                       # D2H: frame offsets + counters only
        else:
            assert up_by_kernel == 0 and cnt[1] >= raw.size
        c.close()
    (a, ao, ac, an), (b, bo, bc, bn) = res[True], res[False]
    assert an == bn == len(lines) + 1 and ac == bc == raw.size and np.array_equal(ao, bo)
    same_lines(a, b, an)
    fake.fg_free_pinned(pb)


def test_a_pinned_raw_stream_is_framed_and_decoded_by_one_launch(fake):
    """round 6: a raw chunk in pinned memory is ONE launch -- the decode kernel frames it itself (fg_fused.hpp) and writes rows, entries
    and frame offsets into pinned memory: no byte of the stream and no table byte crosses by hipMemcpy.  Same result as the pageable
    chunk through the older forms, final or not; a table estimate that was too small and a look-back that gave up both fall back,
    and the next chunk is sized from the first."""
    rng = np.random.default_rng(9)
    lines = corpus(60_000, rng)
    raw = np.frombuffer(b"".join(ln + b"\n" for ln in lines) + b"<1>unterminated tail a=1", np.uint8).copy()
    fake.fg_alloc_pinned.argtypes = [u64, C.POINTER(vp)]
    fake.fg_free_pinned.argtypes = [vp]
    fake.fg_last_host_path.argtypes = [vp]
    pb = vp()
    assert fake.fg_alloc_pinned(raw.size + 64, C.byref(pb)) == 0
    hb = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_uint8)), (raw.size + 64,))
    hb[:] = 0x0A  # (what lies behind the chunk is the caller's: terminators, on purpose)
    hb[:raw.size] = raw
    pad = np.concatenate([raw, np.zeros(64, np.uint8)])
    cnt = (C.c_ulonglong * 3)()

    def call(c, src, final):
        st, po, nf, cons = L.fg_tables(), vp(), u64(), u64()
        fake.fgf_launches(1)
        fake.fgf_counters(cnt, 1)
        assert fake.fg_frame_decode_batch(c.h, 0, 1, src, raw.size, final, C.byref(st), C.byref(po), C.byref(nf), C.byref(cons)) == 0
        fake.fgf_counters(cnt, 1)
        n = int(nf.value)
        offs = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), (n + 1,)).copy()
        return snapshot(st, n), offs, int(cons.value), n, int(fake.fgf_launches(1)), int(fake.fg_last_host_path(c.h)), int(cnt[1]), int(cnt[2])

    for final in (1, 0):
        c = Ctx(fake)
        a, ao, ac, an, launches, path, h2d, d2h = call(c, pb, final)   # one launch, nothing copied but a few counters
        assert path == L.FG_PATH_FRAME_FUSED and launches == 1 and h2d < 4096 and d2h < 4096, (path, launches, h2d, d2h)
        b, bo, bc, bn, _, pathb, h2d_b, _ = call(c, pad.ctypes.data, final)   # pageable: the older forms
        assert pathb != L.FG_PATH_FRAME_FUSED and h2d_b >= raw.size
        assert an == bn == len(lines) + (1 if final else 0) and ac == bc and np.array_equal(ao, bo)
        assert ac == (raw.size if final else raw.size - len(b"<1>unterminated tail a=1"))
        same_lines(a, b, an)
        # a look-back that gives up (bounded wait): the call comes back through the older form with the same result
        fake.fake_fused_abort_next(1)
        d, do, dc, dn, _, pathd, _, _ = call(c, pb, final)
        assert pathd != L.FG_PATH_FRAME_FUSED and dn == an and np.array_equal(do, ao)
        same_lines(a, d, an)
        c.close()
    # a ctx that has seen no chunk sizes its tables for one frame per 200 bytes and one entry per 16: a chunk of three-byte lines with an
    # entry each fits neither -> the launch says what it holds and is repeated with that (once per table); the NEXT chunk is sized from
    # this one: one launch
    short = np.frombuffer(b"a=\n" * 700_000, np.uint8)
    hb[:short.size] = short
    c = Ctx(fake)
    got = []
    for _ in range(2):
        st, po, nf, cons = L.fg_tables(), vp(), u64(), u64()
        fake.fgf_launches(1)
        assert fake.fg_frame_decode_batch(c.h, 0, 1, pb, short.size, 1, C.byref(st), C.byref(po), C.byref(nf), C.byref(cons)) == 0
        got.append((int(nf.value), int(cons.value), int(fake.fg_last_host_path(c.h)), snapshot(st, int(nf.value)), int(fake.fgf_launches(1))))
    assert got[0][:3] == got[1][:3] == (700_000, short.size, L.FG_PATH_FRAME_FUSED)
    assert got[0][4] == 3 and got[1][4] == 2, (got[0][4], got[1][4])  # rows, then entries; then only the entries (one per 3 bytes, not per 16)
    same_lines(got[0][3], got[1][3], 700_000)
    assert got[1][3]["used"] == 700_000
    c.close()
    fake.fg_free_pinned(pb)


def test_a_failed_device_allocation_is_an_error_and_the_ctx_stays_usable(fake):
    rng = np.random.default_rng(4)
    data, offsets = pack(corpus(20_000, rng))
    n = len(offsets) - 1
    c = Ctx(fake)
    st = L.fg_tables()
    fake.fgf_fail_malloc_after(1)
    rc = fake.fg_decode_batch(c.h, 0, data.ctypes.data, data.size, offsets.ctypes.data, n, C.byref(st))
    fake.fgf_fail_malloc_after(-1)
    assert rc == -2  # FG_ERR_HIP
    assert fake.fg_decode_batch(c.h, 0, data.ctypes.data, data.size, offsets.ctypes.data, n, C.byref(st)) == 0
    same_lines(snapshot(st, n), device_reference(fake, data, offsets, data.size // 16 + 1024), n)
    c.close()


def test_argument_errors(fake):
    c = Ctx(fake)
    st = L.fg_tables()
    data, offsets = pack([b"<1>a=1", b"<2>b"])
    assert fake.fg_decode_batch(c.h, 0, data.ctypes.data, data.size, None, 2, C.byref(st)) == -1
    bad = offsets.copy()
    bad[2] = data.size + 5
    assert fake.fg_decode_batch(c.h, 0, data.ctypes.data, data.size, bad.ctypes.data, 2, C.byref(st)) == -1
    assert fake.fg_decode_batch(c.h, 0, data.ctypes.data, data.size, offsets.ctypes.data, 0, C.byref(st)) == 0
    assert fake.fg_decode_batch(c.h, 0, data.ctypes.data, data.size, offsets.ctypes.data, 2, C.byref(st)) == 0
    assert int(snapshot(st, 2)["ent_count"].sum()) == 1
    c.close()


@pytest.mark.parametrize("dense", [False, True])
def test_transcode_sliced_equals_one_piece(fake, dense):
    """fg_transcode_batch from 16 MiB: slices that grow from 4 MiB, one stream per direction of the link (upload, decode, count /
    scan from the running base, write / downloads, the fixed-size arrays once at the end), output buffers sized from the first
    slice; same stream, offsets, row meta and encoder status as the one-piece
    form -- also when the entry table is too small for the sliced form (it then hands the batch to the one-piece form)."""
    fake.fg_transcode_batch.argtypes = [vp, C.c_int, C.c_int, C.POINTER(L.fg_encode_cfg), vp, u64, vp, u64, C.c_int, C.POINTER(L.fg_transcoded)]
    rng = np.random.default_rng(6)
    lines = corpus(300_000, rng, 30 if dense else 0, 40 if dense else 5, 300 if dense else 240)
    data, offsets = pack(lines)
    n = len(lines)
    assert data.size > (64 << 20)
    ecfg = L.fg_encode_cfg()
    ecfg.encoder, ecfg.merger = L.FG_ENC_GELF, L.FG_MERGE_LINE
    res = {}
    for one_piece in (False, True):
        c = Ctx(fake)
        lo = L.fg_launch_opts()
        lo.flags = L.FG_LO_TRANSCODE_ONE_PIECE if one_piece else 0
        assert fake.fg_set_launch_opts(c.h, C.byref(lo)) == 0
        out = L.fg_transcoded()
        fake.fgf_launches(1)
        assert fake.fg_transcode_batch(c.h, 0, 0, C.byref(ecfg), data.ctypes.data, data.size, offsets.ctypes.data, n, 1, C.byref(out)) == 0
        launches = fake.fgf_launches(1)
        assert int(out.n) == n and int(out.consumed) == data.size
        nb = int(out.out_bytes)
        get = lambda p, dt, cnt: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (cnt * np.dtype(dt).itemsize,)).view(dt).copy()  # noqa: E731
        res[one_piece] = (get(out.out, np.uint8, nb), get(out.out_offsets, np.uint64, n + 1), get(out.meta, np.uint32, n), get(out.enc_status, np.uint8, n),
                          launches)
        c.close()
    a, b = res[False], res[True]
    for x, y in zip(a[:4], b[:4]):
        assert np.array_equal(x, y)
    # what the fake encoder writes for an Ok row: the line, one '#' per entry, '\\n'
    ok = (a[2] & 0xFF) == 0
    want = sum(len(ln) + ln.count(b"=") + 1 for ln, k in zip(lines, ok) if k)
    assert int(a[1][-1]) == len(a[0]) == want
    i = int(np.nonzero(ok)[0][12345])
    assert a[0][int(a[1][i]):int(a[1][i + 1])].tobytes() == lines[i] + b"#" * lines[i].count(b"=") + b"\n"
    if not dense:
        assert a[4] > b[4] >= 1  # the sliced form launched one decode per slice


def test_the_timeline_probe_runs_against_the_fake_runtime(fake, tmp_path):
    """tools/probe/e2e_timeline.cpp (the per-slice timeline of the sliced host path, for the GPU box) compiled with g++ against the
    fake runtime: both modes walk their slices, tables and entry ranges without a fault (the times it prints here mean nothing)."""
    from flowgger_amd import synth

    exe = tmp_path / "e2e_timeline_fake"
    subprocess.run(["g++", "-std=c++17", "-O1", "-w", f"-I{HERE / 'fakehip'}", f"-I{ROOT / 'include'}", str(ROOT / "tools/probe/e2e_timeline.cpp"), "-o", str(exe),
                    f"-L{HERE}", "-lhost_pipeline_fake", f"-Wl,-rpath,{HERE}"], check=True)
    corpus_file = tmp_path / "lines.txt"
    corpus_file.write_bytes(b"\n".join(synth.rfc5424_lines(60_000, cfg=4, sd=True)))
    for mode, mib in (("all", "8"), ("collect", "1")):
        r = subprocess.run([str(exe), "rfc5424", str(corpus_file), mib, mode], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-500:]
        assert "total" in r.stdout.splitlines()[-1] and f"mode {mode}" in r.stdout.splitlines()[0]


def test_pinned_pool_reuses_blocks_and_falls_back_to_pageable_memory(fake):
    """fg_alloc_pinned / fg_free_pinned: a freed block is handed to the next request it fits (no pin / unpin per
    connection), idle blocks are bounded, and beyond the cap of pinned bytes a request gets pageable memory that the same free takes."""
    lib = fake
    lib.fg_alloc_pinned.argtypes = [u64, C.POINTER(vp)]
    lib.fg_free_pinned.argtypes = [vp]
    lib.fg_free_pinned.restype = None
    lib.fg_set_pinned_limits.argtypes = [u64, u64]
    lib.fg_pinned_stats.argtypes = [C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]

    def stats():
        a, b, c = u64(), u64(), u64()
        lib.fg_pinned_stats(C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    lib.fg_set_pinned_limits(1 << 30, 0)  # (earlier tests left idle blocks: unpinned here, so that the books start clean)
    p0, i0, l0 = stats()
    assert i0 == 0
    lib.fg_set_pinned_limits(p0 + (8 << 20), 4 << 20)
    a = vp()
    assert lib.fg_alloc_pinned(3 << 20, C.byref(a)) == 0 and a.value
    C.memset(a, 0x5A, 3 << 20)
    assert stats()[0] == p0 + (3 << 20) and stats()[2] == l0 + 1
    lib.fg_free_pinned(a)
    assert stats() == (p0 + (3 << 20), i0 + (3 << 20), l0)          # kept, idle
    b = vp()
    assert lib.fg_alloc_pinned((2 << 20) + 5, C.byref(b)) == 0 and b.value == a.value   # the idle block fits: reused
    c = vp()
    assert lib.fg_alloc_pinned(4 << 20, C.byref(c)) == 0 and c.value                    # a second pinned block: 7 of 8 MiB
    d = vp()
    assert lib.fg_alloc_pinned(4 << 20, C.byref(d)) == 0 and d.value                    # over the cap: pageable, still usable
    C.memset(d, 0x11, 4 << 20)
    assert stats()[0] == p0 + (7 << 20)
    for q in (b, c, d):
        lib.fg_free_pinned(q)
    pinned, idle, live = stats()
    assert live == l0 and idle <= i0 + (4 << 20) and pinned <= p0 + (7 << 20)   # the second free exceeded the idle bound: unpinned
    lib.fg_set_pinned_limits(8 << 30, 256 << 20)
