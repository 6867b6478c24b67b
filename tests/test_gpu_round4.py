"""-m gpu, round 4: parity hardening.

* whole-line mutation fuzz for RFC5424 with structured data (round 3 mutated bytes 0..119 only; the structured data of a 554-byte
  line spans bytes ~70..450) -- rfc5424_decoder.rs:127-242;
* a structural mutation fuzz for LTSV (TAB, ':', '[', ']', digits, signs, multi-byte) -- ltsv_decoder.rs:87-221;
* >= 1 M mutated lines per format, generated on the fly (seeded, vectorised), decoded through fg_decode_batch_device and compared
  with the oracle as canonical Records in chunks -- including lines that straddle the LDS tile end and the 768-byte head-staging
  threshold of the RFC5424 kernel.
The oracle is the checker (tests/oracle_binding.py); the product path is the C ABI.
"""
import numpy as np
import pytest

from flowgger_amd import GelfDecoder, LTSVDecoder, RFC5424Decoder, synth
from gpu_util import assert_same, device_path, host_path_blob

pytestmark = pytest.mark.gpu

RFC5424, LTSV, GELF = 0, 1, 2
IDEO, NBSP = "\u3000", "\u00a0"


@pytest.fixture(scope="module")
def oracle():
    import oracle_binding

    return oracle_binding.Oracle()


def both_paths(dec, oracle, lines, config=None, host=True):
    data, offsets = synth.pack(lines)
    oblob, ooffs = oracle.decode_batch(dec.fmt, data, offsets, config)
    if host:
        (blob, offs), _ = host_path_blob(dec, data, offsets)
        assert_same(blob, offs, oblob, ooffs, lines)
    tables, _, _ = device_path(dec, data, offsets)
    blob2, offs2 = tables.to_host().serialize(dec.fmt, data, offsets, cfg=dec._cfg)
    assert_same(blob2, offs2, oblob, ooffs, lines)


def mutate_py(lines, rng, alphabet, per_line=(1, 4), lo=0):
    """byte-level mutations anywhere in the line (insertions, deletions and multi-byte scalars included; UTF-8 stays valid: only
    ASCII bytes are replaced)"""
    out = []
    for ln in lines:
        b = bytearray(ln)
        for _ in range(int(rng.integers(*per_line))):
            if len(b) <= lo:
                break
            pos = int(rng.integers(lo, len(b)))
            if b[pos] >= 0x80:
                continue
            b[pos:pos + 1] = alphabet[int(rng.integers(0, len(alphabet)))]
        out.append(bytes(b))
    return out


SD_ALPHABET = [b" ", b"[", b"]", b'"', b"\\", b"=", b"<", b">", b"-", b"1", b"Z", b":", b".", b"+", b"\t", b"", b"  ", b'""', b"\\\\",
               b'\\"', b"] ", b"][", b'="', b'" ', "é".encode(), IDEO.encode(), NBSP.encode(), b"\x7f", b"T", b"a"]


def test_rfc5424_sd_whole_line_mutation_fuzz(oracle):
    """Mutations over the WHOLE line -- header, every pair of the structured data, the brackets between elements, the message."""
    rng = np.random.default_rng(0x5424_4)
    dec = RFC5424Decoder()
    base = synth.rfc5424_lines(6000, cfg=4, sd=True, invalid_frac=0)
    lines = mutate_py(base, rng, SD_ALPHABET)
    # the same mutations confined to the structured data (from the first '[' to the last ']')
    for ln in base[:3000]:
        a, z = ln.find(b"["), ln.rfind(b"]")
        b = bytearray(ln)
        for _ in range(int(rng.integers(1, 3))):
            pos = int(rng.integers(a, z + 1))
            if pos < len(b) and b[pos] < 0x80:
                b[pos:pos + 1] = SD_ALPHABET[int(rng.integers(0, len(SD_ALPHABET)))]
        lines.append(bytes(b))
    both_paths(dec, oracle, lines)
    # ... and under the kernel variants the launch options select (head-only staging, few lines per group, small tiles)
    for opts in ({"force_head": True}, {"lines_per_group": 7, "tile_cap": 6144}, {"tile_cap": 4096}):
        d2 = RFC5424Decoder()
        d2.set_launch_opts(**opts)
        both_paths(d2, oracle, lines[:4000], host=False)


@pytest.mark.parametrize("opts", [dict(sd_pairs=True), dict(sd_walk=True), dict(sd_pairs=True, tile_cap=6144, lines_per_group=9),
                                  dict(sd_pairs=True, force_head=True), dict(sd_pairs=True, tile_cap=36864), dict(sd_pairs=True, tile_cap=4096)])
def test_pair_parallel_sd_walk_variants(oracle, opts):
    """The pair-parallel structured-data walk (fg_sd2.hpp, the default from 320-byte lines) and the lane-per-line walker it replaced
    must both give the oracle's Records -- on the structured-data corpus, the short-line corpus (where the walk is forced on), long
    lines, the hand-written shapes of the CPU suite and mutated lines, under several tile geometries."""
    from test_sd2_cpu import HDR

    rng = np.random.default_rng(7)
    dec = RFC5424Decoder()
    dec.set_launch_opts(**opts)
    base = synth.rfc5424_lines(20_000, cfg=4, sd=True)
    lines = base + synth.rfc5424_lines(5000, cfg=2) + synth.rfc5424_lines(3000, cfg=5, sd=True, long_tail=True)
    bodies = [b'[a b="c"] m', b'[a b="c"][d e="f"] m', b'[a b="c"]', b'[a b="c"]x', b'[a b="c"][d ] m', b'[a b="c"c="d"] m', b'[a "b="c"] m', b'[id ] m',
              b'[id] m', b'[a b= "c"] m', b'[a b="c" m', b'[a b="c', b'[a b="c\\"] m', b'[a  b="c"] m', b'[a b="c"  ] m', b'[a b="c"] [d e="f"] m',
              b'[a b="' + b"\\" * 16 + b'"] m', b'[a b="' + b"\\" * 17 + b'" c="d"] m', b'[a ' + b" ".join(b'k%d="v%d"' % (k, k) for k in range(70)) + b'] m',
              b'[a b="c"] "quoted" message "with" quotes', b'[a b="c" d="e] f"] m', b'[nospace]', b'[a b="c"]]', b'[a \tb="c"] m']
    lines += [HDR + b for b in bodies] * 3
    lines += mutate_py(base[:6000], rng, SD_ALPHABET)
    order = rng.permutation(len(lines))
    both_paths(dec, oracle, [lines[i] for i in order], host=False)


LTSV_ALPHABET = [b"\t", b":", b"[", b"]", b"0", b"9", b"-", b"+", b".", b"e", b" ", b"", b"\t\t", b"::", b"time:", b"host:", b"level:", b"\tx",
                 b"\tlevel:7", b"\tlevel:8", b"\tcounter:18446744073709551616", b"\tdone:TRUE", b"\tmean:1e400", b"\tscore:-0",
                 "é".encode(), IDEO.encode(), b"T", b"Z", b"/"]


def test_ltsv_structural_mutation_fuzz(oracle):
    """LTSV has no structure but TAB and the first ':' of a part; typed values, the three timestamp spellings and the bracket strip
    hang off single bytes -- mutate exactly those."""
    rng = np.random.default_rng(0x175)
    dec = LTSVDecoder(synth.LTSV_CONFIG)
    base = synth.ltsv_lines(8000, invalid_frac=0) + synth.ltsv_lines(1500, invalid_frac=0, long_tail=True)
    lines = mutate_py(base, rng, LTSV_ALPHABET)
    both_paths(dec, oracle, lines, synth.LTSV_CONFIG)
    d2 = LTSVDecoder(synth.LTSV_CONFIG)
    d2.set_launch_opts(lines_per_group=5, tile_cap=4096)
    both_paths(d2, oracle, lines[:4000], synth.LTSV_CONFIG, host=False)


@pytest.mark.parametrize("opts", [dict(), dict(force_head=True), dict(no_head=True), dict(force_head=True, tile_cap=6144, lines_per_group=5)])
def test_ltsv_long_lines_head_staging(oracle, opts):
    """LTSV lines of 64 B .. 8 KiB: only the head of a long line is staged, what lies behind it is scanned for a TAB without being
    stored (fg_ltsv.hip LtsvFormatT<true>).  Lines whose long part is the last one (the corpus), lines with MORE parts behind the
    first KiB (typed ones, `time` / `host` / `level` back there, parts without ':'), a cut part that is typed, a cut name, TABs exactly
    at the head's end -- all must give the oracle's Records, with head staging forced on, off, and left to the launcher."""
    rng = np.random.default_rng(11)
    dec = LTSVDecoder(synth.LTSV_CONFIG)
    dec.set_launch_opts(**opts)
    base = synth.ltsv_lines(6000, invalid_frac=0.005, long_tail=True)
    pad = lambda k: b"x" * int(k)  # noqa: E731
    extra = []
    for i in range(1500):
        k = int(rng.integers(900, 1200))  # (around the 1 KiB head, every alignment)
        head = b"time:1.5\thost:h\tmessage:" + pad(k)
        tails = [b"\tlevel:3", b"\tlevel:9", b"\tcounter:12", b"\tcounter:x", b"\tnovalue", b"\thost:late", b"\ttime:[2015-08-05T15:53:45Z]", b"\t", b"",
                 b"\tmean:1e3\tdone:true", b"\tk:" + pad(int(rng.integers(1, 3000))), b"\tscore:-" + b"9" * int(rng.integers(1, 30))]
        extra.append(head + tails[i % len(tails)])
        # a cut part that is not a plain string: the typed value / the timestamp / the name itself straddles the head's end
        extra.append(b"host:h\ttime:1\tpad:" + pad(k - 40) + b"\tcounter:" + b"7" * int(rng.integers(1, 19)))
        extra.append(b"host:h\tpad:" + pad(k - 30) + b"\ttime:" + b"1" * int(rng.integers(1, 15)) + b".5")
        extra.append(b"host:h\ttime:2\tpad:" + pad(k - 35) + b"\t" + b"n" * int(rng.integers(1, 60)) + b":v")
    lines = base + mutate_py(base[:2000], rng, LTSV_ALPHABET) + extra
    order = rng.permutation(len(lines))
    both_paths(dec, oracle, [lines[i] for i in order], synth.LTSV_CONFIG, host=False)


def _mutate_packed(data, offsets, rng, ascii_alphabet, frac=0.85):
    """vectorised: one to three single-byte replacements in `frac` of the lines (ASCII for ASCII: offsets and UTF-8 validity stay)"""
    data = data.copy()
    n = len(offsets) - 1
    ln = np.diff(offsets.astype(np.int64))
    alpha = np.frombuffer(ascii_alphabet, np.uint8)
    for _ in range(3):
        pick = (rng.random(n) < frac / 2) & (ln > 0)
        idx = np.nonzero(pick)[0]
        pos = offsets[idx].astype(np.int64) + (rng.integers(0, 1 << 62, idx.size) % ln[idx])
        ok = data[pos] < 0x80
        data[pos[ok]] = alpha[rng.integers(0, alpha.size, int(ok.sum()))]
    return data


def _chunked_compare(dec, oracle, data, offsets, config, chunk=250_000):
    n = len(offsets) - 1
    for i0 in range(0, n, chunk):
        i1 = min(n, i0 + chunk)
        b0, b1 = int(offsets[i0]), int(offsets[i1])
        d = np.ascontiguousarray(data[b0:b1])
        o = (offsets[i0:i1 + 1] - offsets[i0]).astype(np.uint64)
        oblob, ooffs = oracle.decode_batch(dec.fmt, d, o, config)
        tables, _, _ = device_path(dec, d, o)
        blob, offs = tables.to_host().serialize(dec.fmt, d, o, cfg=dec._cfg)
        if not (np.array_equal(offs, ooffs) and np.array_equal(blob, oblob)):
            lines = [bytes(d[int(o[k]):int(o[k + 1])]) for k in range(i1 - i0)]
            assert_same(blob, offs, oblob, ooffs, lines)


@pytest.mark.parametrize("fmt", ["rfc5424_sd", "ltsv", "gelf"])
def test_one_million_mutated_lines_per_format(oracle, fmt):
    """>= 1 M mutated lines, seeded, generated on the fly: a 125 K-line tile (with lengths that straddle the LDS tile end and the
    768-byte head threshold mixed in) mutated eight different ways; every chunk == the oracle, Record for Record."""
    rng = np.random.default_rng({"rfc5424_sd": 41, "ltsv": 42, "gelf": 43}[fmt])
    if fmt == "rfc5424_sd":
        dec, config = RFC5424Decoder(), None
        base = synth.rfc5424_lines(100_000, cfg=4, sd=True, invalid_frac=0.002)
        # lines around the head-staging threshold (768 B average switches kernels; 1024 B is the staged head) and around the tile end
        base += synth.rfc5424_lines(15_000, cfg=5, sd=True, invalid_frac=0, long_tail=True)
        base += [ln + b" pad" * int(k) for ln, k in zip(synth.rfc5424_lines(10_000, cfg=4, sd=True, invalid_frac=0), rng.integers(40, 160, 10_000))]
        alpha = b' []"\\=<>-1Z:.+a\x7f'
    elif fmt == "ltsv":
        dec, config = LTSVDecoder(synth.LTSV_CONFIG), synth.LTSV_CONFIG
        base = synth.ltsv_lines(110_000, invalid_frac=0.002) + synth.ltsv_lines(15_000, invalid_frac=0, long_tail=True)
        alpha = b"\t:[]09-+.e TZ/a"
    else:
        dec, config = GelfDecoder(), None
        base = synth.gelf_lines(125_000, invalid_frac=0.002)
        alpha = b'{}[],:"\\ 019-+.eEtfn a_'
    order = rng.permutation(len(base))  # (long and short lines interleaved: groups are cut by bytes)
    base = [base[i] for i in order]
    data, offsets = synth.pack(base)
    total = 0
    for rep in range(8):
        d = _mutate_packed(data, offsets, rng, alpha)
        _chunked_compare(dec, oracle, d, offsets, config)
        total += len(base)
    assert total >= 1_000_000


@pytest.mark.parametrize("fmt", ["rfc5424_sd", "ltsv", "gelf"])
def test_zero_copy_decode_batch_from_pinned_buffers(oracle, fmt):
    """fg_decode_batch with bytes + offsets in pinned memory takes the ZERO-COPY form (one launch, the kernels read the lines over the
    link and write the table columns into pinned host memory): the Records must be the oracle's, the same as through the sliced copies
    (FG_LO_NO_ZERO_COPY) -- also when the first entry-table size is too small and the launch is repeated."""
    import ctypes as C

    from flowgger_amd import _lib as L
    from flowgger_amd.tables import HostTables

    if fmt == "rfc5424_sd":
        dec, config = RFC5424Decoder(), None
        lines = synth.rfc5424_lines(60_000, cfg=4, sd=True) + synth.rfc5424_lines(4000, cfg=5, sd=True, long_tail=True)
        lines += [b'<13>1 2015-08-05T15:53:45Z h a p m [x ' + b" ".join(b'k%d="v"' % k for k in range(60)) + b"] dense pairs"] * 3000  # > 1 entry / 16 B
    elif fmt == "ltsv":
        dec, config, lines = LTSVDecoder(synth.LTSV_CONFIG), synth.LTSV_CONFIG, synth.ltsv_lines(60_000) + synth.ltsv_lines(3000, long_tail=True)
    else:
        dec, config, lines = GelfDecoder(), None, synth.gelf_lines(60_000)
    data, offsets = synth.pack(lines)
    n = len(lines)
    lib = L.lib()
    pb, po = C.c_void_p(), C.c_void_p()
    L.check(lib.fg_alloc_pinned(data.size + 64, C.byref(pb)), "fg_alloc_pinned")
    L.check(lib.fg_alloc_pinned((n + 1) * 8, C.byref(po)), "fg_alloc_pinned")
    try:
        hb = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_uint8)), (data.size + 64,))
        ho = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), (n + 1,))
        hb[:data.size] = data
        hb[data.size:] = 0
        ho[:] = offsets
        oblob, ooffs = oracle.decode_batch(dec.fmt, data, offsets, config)
        for no_zc in (False, True):
            dec.set_launch_opts(no_zero_copy=no_zc)
            st = L.fg_tables()
            L.check(lib.fg_decode_batch(dec._ctx, dec.fmt, pb, data.size, po, n, C.byref(st)), "fg_decode_batch")
            blob, offs = HostTables.from_struct(st).serialize(dec.fmt, data, offsets, cfg=dec._cfg)
            assert_same(blob, offs, oblob, ooffs, lines)
    finally:
        lib.fg_free_pinned(pb)
        lib.fg_free_pinned(po)



@pytest.mark.parametrize("framing", ["line", "nul"])
def test_one_pass_framing_equals_the_three_kernel_form_at_scale(framing):
    """k_frame_onepass (a chained scan with decoupled look-back over one descriptor per 16 KiB tile) against the classic scan -> prefix ->
    emit (FG_LO_FRAME_CLASSIC) on a stream of ~20 000 tiles with injected UTF-8 damage, empty frames, lines longer than a tile and an
    unterminated tail: identical offsets and verdicts -- and the oracle's splitter on a sample.  Also through fg_frame_decode_batch
    (sliced: every slice continues the chain where the one before stopped)."""
    import ctypes as C

    import torch
    from flowgger_amd import _lib as L

    rng = np.random.default_rng(0x0F4A)
    dec = RFC5424Decoder()
    dev = torch.device("cuda", dec.device)
    delim = b"\n" if framing == "line" else b"\0"
    base = synth.rfc5424_lines(120_000, cfg=2)
    dmg = [b"\xff", b"\xc3", b"\xe2\x82", b"\xf0\x9f\x98", b"\xed\xa0\x80", b"\xc0\xaf", b"caf\xc3\xa9 \xe2\x82\xac ok"]
    pieces = []
    for i, ln in enumerate(base):
        if i % 53 == 7:
            ln = ln + b" " + dmg[(i // 53) % len(dmg)]
        if i % 997 == 0:
            ln = b""
        if i % 20011 == 5:
            ln = ln + b" " + b"x" * int(rng.integers(17_000, 70_000))  # longer than a tile (and than four)
        pieces.append(ln + delim)
    one = b"".join(pieces)
    raw = one * 10 + b"unterminated tail \xe2\x82"
    assert len(raw) > 256 << 20
    d_bytes = torch.cat([torch.frombuffer(bytearray(raw), dtype=torch.uint8), torch.zeros(32, dtype=torch.uint8)]).to(dev)
    d_raw = d_bytes[:len(raw)]
    fr = L.FG_FRAME_LINE if framing == "line" else L.FG_FRAME_NUL
    res = {}
    for classic in (False, True):
        dec.set_launch_opts(frame_classic=classic)
        d_offsets, d_bad, n = dec.frame_device(d_raw, fr)
        res[classic] = (d_offsets[:n + 1].cpu().numpy().copy(), d_bad[:n].cpu().numpy().copy(), n)
    (ao, ab, an), (bo, bb, bn) = res[False], res[True]
    assert an == bn == len(pieces) * 10 + 1
    assert np.array_equal(ao, bo) and np.array_equal(ab, bb)
    assert int(ao[-1]) == len(raw) and int(ab.sum()) > 10 * (len(base) // 53) * 5 // 7
    # the first copy of the stream against the plain-Python splitter
    ends = np.cumsum(np.fromiter((len(p) for p in pieces), np.int64, len(pieces)))
    assert np.array_equal(ao[1:len(pieces) + 1].astype(np.int64), ends)
    for i in range(0, len(pieces), 37):
        body = pieces[i][:-1]
        try:
            body.decode("utf-8")
            ok = 1
        except UnicodeDecodeError:
            ok = 0
        assert int(ab[i]) == 1 - ok, i
    # ... and the sliced host path (pageable chunk: hipMemcpy uploads, every slice framed by one launch of the chain)
    lib = L.lib()
    host = np.frombuffer(raw, np.uint8)
    pad = np.concatenate([host, np.zeros(64, np.uint8)])
    out = {}
    for classic in (False, True):
        dec.set_launch_opts(frame_classic=classic)
        st, po, nf, cons = L.fg_tables(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        L.check(lib.fg_frame_decode_batch(dec._ctx, dec.fmt, fr, pad.ctypes.data, host.size, 1, C.byref(st), C.byref(po), C.byref(nf), C.byref(cons)),
                "fg_frame_decode_batch")
        n = int(nf.value)
        offs = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), (n + 1,)).copy()
        meta = np.ctypeslib.as_array(C.cast(st.meta, C.POINTER(C.c_uint32)), (n,)).copy()
        out[classic] = (offs, meta, n)
    assert out[False][2] == out[True][2] == an
    assert np.array_equal(out[False][0], out[True][0]) and np.array_equal(out[False][0].astype(np.int64), ao.astype(np.int64))
    assert np.array_equal(out[False][1], out[True][1])
    assert np.array_equal((out[False][1] & 0xFF) == L.FG_ST_BAD_UTF8, ab == 1)
    dec.set_launch_opts()


def test_device_merge_of_tagged_sub_batches_equals_the_host_merge():
    """fg_merge_tables_device: the rows of the two sub-batches of a mixed stream (configs[4]) go back to their arrival positions while the
    tables are still in HBM -- the merged table, copied out once, must equal what fg_merge_tables makes of the two tables on the host
    (rows at their positions, entries rebased and behind one another, src_part = the tag)."""
    import torch
    from flowgger_amd import shard

    n = 200_000
    tag, (la, ia), (lb, ib) = synth.mixed_cfg5(n)
    decs = (RFC5424Decoder(), LTSVDecoder(synth.LTSV_CONFIG))
    dev = torch.device("cuda", decs[0].device)
    dparts, hparts, index = [], [], [ia, ib]
    for dec, lines in zip(decs, (la, lb)):
        data, offsets = synth.pack(lines)
        tables, _, _ = device_path(dec, data, offsets, ent_cap=int(offsets[-1]) // 16 + (1 << 20))
        dparts.append(tables)
        hparts.append(tables.to_host_pinned())
    want, wsrc = shard.merge_tables(hparts, index)
    d_index = [torch.from_numpy(ix.astype(np.int64)).to(dev) for ix in index]
    out, d_src = shard.merge_tables_device(decs[0], dparts, d_index)
    torch.cuda.synchronize(dev)
    got = out.to_host_pinned()
    assert got.n == want.n == n
    assert np.array_equal(d_src.cpu().numpy(), tag) and np.array_equal(wsrc, tag)
    for col in ("meta", "ts", "hostname", "appname", "procid", "msgid", "msg", "full_msg", "ent_count"):
        assert np.array_equal(got.a[col][: n * (2 if got.a[col].size >= 2 * n else 1)], want.a[col][: n * (2 if want.a[col].size >= 2 * n else 1)]), col
    # round 5: the device merge lays the entries out DENSE and in arrival order (the host merge keeps each part's layout, stranded
    # reservations included): same entries per row, different places
    cnt = want.a["ent_count"][:n].astype(np.int64)
    assert got.ent_used == int(cnt.sum()) <= want.ent_used
    dense = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    assert np.array_equal(got.a["ent_first"][:n][cnt > 0].astype(np.int64), dense[cnt > 0])
    take = np.repeat(want.a["ent_first"][:n].astype(np.int64) - dense, cnt) + np.arange(int(cnt.sum()))
    for col, w in (("ent_name", 2), ("ent_val", 1), ("ent_type", 1), ("ent_flags", 1)):
        g_ = got.a[col][: got.ent_used * w].reshape(-1, w)
        w_ = want.a[col][: want.ent_used * w].reshape(-1, w)[take]
        assert np.array_equal(g_, w_), col
    # a second merge into the same output (a framer merges batch after batch into the same memory)
    out2, _ = shard.merge_tables_device(decs[0], dparts, d_index, out=out, d_src=d_src)
    torch.cuda.synchronize(dev)
    assert out2 is out and out.to_host_pinned().ent_used == int(cnt.sum())
