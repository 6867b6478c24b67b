"""Round 6 GPU tests: FRAMING INSIDE THE DECODE KERNELS (flowgger_amd/csrc/fg_fused.hpp, fg_fuse.hpp; fg_frame_decode_device and the
one-launch form of fg_frame_decode_batch) against the oracle -- its restatement of the splitters (fgo_frame; src/flowgger/splitter/
line_splitter.rs:17-25, nul_splitter.rs:18-40) for the frames and the UTF-8 verdicts, its decoders for every frame's Record."""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

from flowgger_amd import GelfDecoder, LTSVDecoder, RFC5424Decoder, synth
from flowgger_amd import _lib as L
from flowgger_amd.tables import DeviceTables, HostTables
from oracle_binding import Oracle

pytestmark = pytest.mark.gpu

RFC5424, LTSV, GELF = 0, 1, 2


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


def make_decoder(fmt):
    if fmt == RFC5424:
        return RFC5424Decoder(), None
    if fmt == LTSV:
        return LTSVDecoder(synth.LTSV_CONFIG), synth.LTSV_CONFIG
    return GelfDecoder(), None


def corpus(fmt, n, sd=False):
    if fmt == RFC5424:
        return synth.rfc5424_lines(n, cfg=4, sd=True) if sd else synth.rfc5424_lines(n, cfg=2)
    if fmt == LTSV:
        return synth.ltsv_lines(n)
    return synth.gelf_lines(n)


def stream_of(lines, framing, crlf_every=0, empty_every=0, damage_every=0, tail=b""):
    term = b"\n" if framing == L.FG_FRAME_LINE else b"\0"
    out = bytearray()
    for i, ln in enumerate(lines):
        if framing == L.FG_FRAME_NUL:
            ln = ln.replace(b"\0", b" ")
        if damage_every and i % damage_every == damage_every - 1:
            ln = ln[: len(ln) // 2] + bytes([(0x80, 0xC3, 0xE2, 0xF0, 0xFF)[i % 5]]) + ln[len(ln) // 2:]
        out += ln
        if crlf_every and framing == L.FG_FRAME_LINE and i % crlf_every == 0:
            out += b"\r"
        out += term
        if empty_every and i % empty_every == 0:
            out += term
    return bytes(out + tail)


def expected_frames(oracle, raw, framing, final):
    starts, ends, valid = oracle.frame_arrays(np.frombuffer(raw + b"\0", np.uint8)[: len(raw)], "line" if framing == L.FG_FRAME_LINE else "nul")
    term = 0x0A if framing == L.FG_FRAME_LINE else 0
    if not final and len(ends) and int(ends[-1]) == len(raw) and (len(raw) == 0 or raw[-1] != term):
        starts, ends, valid = starts[:-1], ends[:-1], valid[:-1]
    return starts, ends, valid


def strip(raw, s, e, framing):
    body = raw[s:e]
    if framing == L.FG_FRAME_LINE:
        if body.endswith(b"\n"):
            body = body[:-1]
            if body.endswith(b"\r"):
                body = body[:-1]
    elif body.endswith(b"\0"):
        body = body[:-1]
    return body


def check_rows(oracle, fmt, cfg, dec_cfg, raw, framing, starts, ends, valid, tab: HostTables, offs, i0=0):
    """rows [i0, n) of `tab` (frame i = raw[offs[i] .. offs[i + 1])) against the oracle's decode of every valid frame"""
    n = len(starts)
    assert np.array_equal(offs[:n].astype(np.uint64), starts), f"frame starts differ at {int(np.flatnonzero(offs[:n] != starts)[0])}"
    assert n == 0 or int(offs[n]) == int(ends[-1])
    if n == i0:
        return
    data = np.frombuffer(raw + b"\0" * 32, np.uint8)
    blob, boffs = tab.serialize(fmt, data, np.ascontiguousarray(offs[: n + 1], np.uint64), i0, n, cfg=dec_cfg)
    st = tab.status[i0:n]
    assert np.array_equal(st == L.FG_ST_BAD_UTF8, valid[i0:] == 0), "UTF-8 verdicts differ"
    good = [strip(raw, int(starts[i]), int(ends[i]), framing) for i in range(i0, n) if valid[i]]
    if not good:
        return
    gdata, goffs = synth.pack(good)
    oblob, ooffs = oracle.decode_batch(fmt, gdata, goffs, cfg)
    j = 0
    for i in range(i0, n):
        if not valid[i]:
            continue
        got = blob[int(boffs[i - i0]):int(boffs[i - i0 + 1])].tobytes()
        want = oblob[int(ooffs[j]):int(ooffs[j + 1])].tobytes()
        assert got == want, f"frame {i}: {good[j][:80]!r}\n  gpu    {got[:120]!r}\n  oracle {want[:120]!r}"
        j += 1


def run_device(dec, raw, framing, final, avg_line=0, cap=None):
    import torch

    dev = torch.device("cuda", dec.device)
    padded = (len(raw) + 15) & ~15
    host = np.full(padded + 16, 0x0A if framing == L.FG_FRAME_LINE else 0, np.uint8)  # what lies behind the stream is terminators on purpose
    host[: len(raw)] = np.frombuffer(raw, np.uint8)
    d_bytes = torch.from_numpy(host).to(dev)[: len(raw)]
    cap = cap if cap is not None else len(raw) // 2 + 16
    tables = DeviceTables(cap, len(raw) // 4 + 4096, dev)
    d_off, d_res = dec.frame_decode_device(d_bytes, framing, tables, cap, final=final, avg_line=avg_line)
    torch.cuda.synchronize(dev)
    res = d_res.cpu().numpy()
    assert int(res[1]) == 0, "the fused launch gave up"
    n = int(res[0])
    return tables, d_off.cpu().numpy().astype(np.uint64), n


@pytest.mark.parametrize("fmt,sd", [(RFC5424, False), (RFC5424, True), (LTSV, False), (GELF, False)])
@pytest.mark.parametrize("framing", [L.FG_FRAME_LINE, L.FG_FRAME_NUL])
def test_fused_frame_decode_matches_the_oracle(oracle, fmt, sd, framing):
    dec, cfg = make_decoder(fmt)
    lines = corpus(fmt, 6000, sd)
    for final, tail in ((True, b""), (True, b"an unterminated last piece"), (False, b"carried over \xe2\x82")):
        raw = stream_of(lines, framing, crlf_every=7, empty_every=501, damage_every=97, tail=tail)
        starts, ends, valid = expected_frames(oracle, raw, framing, final)
        tables, offs, n = run_device(dec, raw, framing, final)
        assert n == len(starts)
        check_rows(oracle, fmt, cfg, dec._cfg, raw, framing, starts, ends, valid, tables.to_host(), offs)


@pytest.mark.parametrize("fmt", [RFC5424, LTSV, GELF])
def test_fused_geometries_that_do_not_fit_the_corpus(oracle, fmt):
    """the launch planned for the wrong line length (tiles of many short lines: several lists and passes; tiles shorter than a line:
    the forward scan and lines parsed from global memory), explicit tile / group overrides, tiny streams"""
    dec, cfg = make_decoder(fmt)
    lines = corpus(fmt, 3000)
    raw = stream_of(lines, L.FG_FRAME_LINE, crlf_every=5, empty_every=3, damage_every=61)
    starts, ends, valid = expected_frames(oracle, raw, L.FG_FRAME_LINE, True)
    for avg in (16, 40, 700):
        tables, offs, n = run_device(dec, raw, L.FG_FRAME_LINE, True, avg_line=avg)
        assert n == len(starts), f"avg_line {avg}"
        check_rows(oracle, fmt, cfg, dec._cfg, raw, L.FG_FRAME_LINE, starts, ends, valid, tables.to_host(), offs)
    for opts in (dict(lines_per_group=3), dict(tile_cap=4096), dict(waves_per_cu=1), dict(lines_per_group=64, tile_cap=32768)):
        dec.set_launch_opts(**opts)
        tables, offs, n = run_device(dec, raw, L.FG_FRAME_LINE, True)
        assert n == len(starts), str(opts)
        check_rows(oracle, fmt, cfg, dec._cfg, raw, L.FG_FRAME_LINE, starts, ends, valid, tables.to_host(), offs)
    dec.set_launch_opts()
    for small in (lines[0] + b"\n", lines[0], b"\n", b"\n\n\n", lines[1] + b"\n" + lines[2]):
        for final in (True, False):
            starts, ends, valid = expected_frames(oracle, small, L.FG_FRAME_LINE, final)
            tables, offs, n = run_device(dec, small, L.FG_FRAME_LINE, final)
            assert n == len(starts), (small[:20], final)
            check_rows(oracle, fmt, cfg, dec._cfg, small, L.FG_FRAME_LINE, starts, ends, valid, tables.to_host(), offs)


def test_fused_many_tiles_row_indices(oracle):
    """64 MB = several thousand tiles and dozens of look-back blocks: every frame start against the oracle's, the Ok count, and the rows
    of a sample of the stream's far end record for record"""
    dec, cfg = make_decoder(RFC5424)
    lines = synth.rfc5424_lines(40_000, cfg=2)
    raw = stream_of(lines, L.FG_FRAME_LINE, crlf_every=11) * 6
    starts, ends, valid = expected_frames(oracle, raw, L.FG_FRAME_LINE, True)
    tables, offs, n = run_device(dec, raw, L.FG_FRAME_LINE, True, cap=len(starts) + 64)
    assert n == len(starts)
    assert np.array_equal(offs[:n], starts) and int(offs[n]) == len(raw)
    host = tables.to_host()
    one = len(lines)
    st = host.status[:n]
    assert np.array_equal(st[:one], st[n - one:n])  # the replicas decode alike
    check_rows(oracle, RFC5424, cfg, dec._cfg, raw, L.FG_FRAME_LINE, starts, ends, valid, host, offs, i0=n - 3000)


def test_fused_table_too_small_reports_the_frame_count(oracle):
    dec, cfg = make_decoder(RFC5424)
    raw = stream_of(synth.rfc5424_lines(5000, cfg=2), L.FG_FRAME_LINE)
    tables, offs, n = run_device(dec, raw, L.FG_FRAME_LINE, True, cap=1000)
    assert n == 5000  # the rows beyond the capacity were not written; the caller runs again with more
    starts, ends, valid = expected_frames(oracle, raw, L.FG_FRAME_LINE, True)
    assert np.array_equal(offs[:1000], starts[:1000])


@pytest.mark.parametrize("fmt", [RFC5424, LTSV, GELF])
def test_frame_decode_batch_one_launch_from_pinned_memory(oracle, fmt):
    """fg_frame_decode_batch with the raw chunk in PINNED memory: the one-launch form (the kernels read the chunk over the link and
    write tables and offsets into pinned memory) against the oracle and against the sliced path of rounds 3-5"""
    dec, cfg = make_decoder(fmt)
    lib = L.lib()
    lines = corpus(fmt, 20_000)
    raw = stream_of(lines, L.FG_FRAME_LINE, crlf_every=9, empty_every=1001, damage_every=89, tail=b"tail without a terminator")
    p = C.c_void_p()
    L.check(lib.fg_alloc_pinned(len(raw) + 64, C.byref(p)), "fg_alloc_pinned")
    try:
        buf = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (len(raw) + 64,))
        buf[:] = 0x0A
        buf[: len(raw)] = np.frombuffer(raw, np.uint8)
        for final in (True, False):
            starts, ends, valid = expected_frames(oracle, raw, L.FG_FRAME_LINE, final)
            for fused in (True, False):
                dec.set_launch_opts(no_fused_framing=not fused)
                st = L.fg_tables()
                off = C.c_void_p()
                nf, used = C.c_uint64(), C.c_uint64()
                L.check(lib.fg_frame_decode_batch(dec._ctx, fmt, L.FG_FRAME_LINE, p, len(raw), int(final), C.byref(st), C.byref(off), C.byref(nf),
                                                  C.byref(used)), "fg_frame_decode_batch")
                n = int(nf.value)
                assert n == len(starts), (final, fused)
                assert int(used.value) == (int(ends[-1]) if n else 0)
                offs = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_uint64)), (n + 1,)).copy()
                check_rows(oracle, fmt, cfg, dec._cfg, raw, L.FG_FRAME_LINE, starts, ends, valid, HostTables.from_struct(st), offs)
    finally:
        dec.set_launch_opts()
        lib.fg_free_pinned(p)


# ---------------------------------------------------------------------------------------------
# RFC3164: lines regrouped by shape (fg_rfc3164.hip; rfc3164_decoder.rs:31-213)
# ---------------------------------------------------------------------------------------------
def test_rfc3164_regrouped_by_shape_matches_the_oracle(oracle):
    """the slow shapes handed to the second kernel (shape key from a line's first bytes, per-class lists, rows staged line by line): every
    Record against the oracle, at batch sizes that do and do not fill a workgroup and a list, with empty / one-byte / very long lines, lines whose
    first bytes lie at every alignment, and the shape key's own edge cases (a capital after the time that is no zone, a year that is
    none); and against the plain kernel on the same batch"""
    from flowgger_amd import RFC3164Decoder, tzdb
    from test_gpu_parity import RFC3164_CONFIG, RFC3164_YEAR
    from test_rfc3164_cpu import fuzz_lines
    from gpu_util import assert_same, device_path

    oracle.set_rfc3164(RFC3164_YEAR, tzdb.default_table())
    dec = RFC3164Decoder(RFC3164_CONFIG)
    base = synth.rfc3164_lines(20_000) + fuzz_lines(6_000, 31)
    base += [b"", b"<", b"<1", b"<12>", b"A", b"Aug", b"Aug 6 11:15:24 Host-With-Capital app: m", b"<34>2019 Aug 6 11:15:24 UTC h a: m", b"<5>1234x", b"a: b: ",
             b"Aug 6 11:15:24 " + b"h" * 9000 + b" long hostname token", b"<190>Oct  1 00:00:00 Europe/Paris h app[1]: " + b"x" * 70_000]
    for n in (1, 63, 64, 65, 1023, 1025, len(base)):
        lines = base[-n:] if n <= 65 else base[:n]
        data, offsets = synth.pack(lines)
        oblob, ooffs = oracle.decode_batch(dec.fmt, data, offsets, None)
        got = {}
        for mode in (1, 2):  # regrouped, plain
            dec.set_launch_opts(rfc3164_regroup=mode)
            tables, _, _ = device_path(dec, data, offsets)
            blob, offs = tables.to_host().serialize(dec.fmt, data, offsets, cfg=dec._cfg)
            assert_same(blob, offs, oblob, ooffs, lines)
            got[mode] = tables.to_host().a["meta"][:len(lines)].copy()
        assert np.array_equal(got[1], got[2])
    dec.set_launch_opts()


def test_rfc3164_regrouped_frames_and_large_batch(oracle):
    """the library's own choice (regrouping from 1 M lines on) on a 1.2 M-line stream of FRAMES (terminators stripped in-kernel, a
    frame that is not valid UTF-8): the Ok verdicts of every row against the plain kernel, a sample of Records against the oracle"""
    import torch

    from flowgger_amd import RFC3164Decoder, tzdb
    from test_gpu_parity import RFC3164_CONFIG, RFC3164_YEAR

    oracle.set_rfc3164(RFC3164_YEAR, tzdb.default_table())
    dec = RFC3164Decoder(RFC3164_CONFIG)
    lines = synth.rfc3164_lines(100_000) * 12
    lines[7] = lines[7] + b"\r"
    lines[9] = b"Aug  6 11:15:24 h \xff\xfe not utf-8"
    stream = b"".join(ln + b"\n" for ln in lines)
    raw = torch.frombuffer(bytearray(stream + b"\0" * 32), dtype=torch.uint8).cuda()[:len(stream)]
    d_off, d_bad, nf = dec.frame_device(raw, L.FG_FRAME_LINE)
    assert nf == len(lines)
    metas = {}
    for mode in (0, 2):
        dec.set_launch_opts(rfc3164_regroup=mode)
        tables = DeviceTables(nf, 16, raw.device)
        dec.decode_frames_device(raw, d_off, nf, tables, L.FG_FRAME_LINE, d_bad)
        torch.cuda.synchronize()
        host = tables.to_host()
        metas[mode] = (host.a["meta"][:nf].copy(), host.a["ts"][:nf].copy(), host.a["hostname"][: 2 * nf].copy(), host.a["msg"][: 2 * nf].copy())
        if mode == 0:
            off = d_off[:nf + 1].cpu().numpy().astype(np.uint64)
            data = np.frombuffer(stream + b"\0" * 32, np.uint8)
            blob, offs = host.serialize(dec.fmt, data, off, 0, 3000)
            good = [ln[:-1] if ln.endswith(b"\r") else ln for ln in lines[:3000]]
            gdata, goffs = synth.pack(good)
            oblob, ooffs = oracle.decode_batch(dec.fmt, gdata, goffs)
            for i in range(3000):
                if i == 9:
                    assert host.status[i] == L.FG_ST_BAD_UTF8
                    continue
                assert blob[int(offs[i]):int(offs[i + 1])].tobytes() == oblob[int(ooffs[i]):int(ooffs[i + 1])].tobytes(), i
    dec.set_launch_opts()
    for a, b in zip(metas[0], metas[2]):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("n_lines", [70_000, 101_003, 170_000])
def test_transcode_growing_slices_equal_one_piece_and_the_oracle(oracle, n_lines):
    """fg_transcode_batch from 16 MiB (round 6): slices of 4, 8, 16 ... MiB on one stream per direction of the link, a short last slice
    merged into the one before, the fixed-size arrays downloaded once at the end.  18 / 26 / 43 MB of cfg2 lines: the same bytes,
    offsets and statuses as the one-piece path, twice (buffers at size the second time), and as the oracle's decode -> encode -> merger
    on a prefix."""
    import oracle_binding as OB
    from flowgger_amd import GelfEncoder, Pipeline

    dec = RFC5424Decoder()
    lines = synth.rfc5424_lines(n_lines, cfg=2, invalid_frac=0.01)
    data, offsets = synth.pack(lines)
    assert data.size > (16 << 20)
    data = np.concatenate([data, np.zeros(64, np.uint8)])
    now_ts = 1438859724.638
    pipe = Pipeline(dec, GelfEncoder(None, merger="line"))
    sliced = pipe.run_packed(data, offsets, now_ts=now_ts)
    again = pipe.run_packed(data, offsets, now_ts=now_ts)
    dec.set_launch_opts(transcode_one_piece=True)
    whole = pipe.run_packed(data, offsets, now_ts=now_ts)
    dec.set_launch_opts()
    for r in (sliced, again):
        assert r.n == len(lines) and r.consumed == int(offsets[-1])
        assert np.array_equal(r.out_offsets, whole.out_offsets) and np.array_equal(r.out, whole.out)
        assert np.array_equal(r.enc_status, whole.enc_status) and np.array_equal(r.dec_status, whole.dec_status)
    m = 20_000
    oblob, ooffs, ost = oracle.decode_encode_batch(RFC5424, OB.ENC_GELF, OB.MERGE_LINE, data, offsets[: m + 1], None, extra=None, prepend=None, now_ts=now_ts)
    assert np.array_equal(sliced.out_offsets[: m + 1], ooffs) and np.array_equal(sliced.out[: int(ooffs[-1])], oblob)
    assert np.array_equal(np.minimum(sliced.enc_status[:m], 2), ost)
