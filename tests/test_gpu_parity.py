"""GPU: parity of the HIP decode path (through the C ABI) against the CPU oracle.

Bit-exact is the bar: the canonical Record serialisation of every line (Ok and Err alike, f64
timestamps as raw IEEE bits) must be byte-identical to the oracle's."""
import numpy as np
import pytest

from flowgger_amd import GelfDecoder, LTSVDecoder, RFC3164Decoder, RFC5424Decoder, pack_lines, synth
from flowgger_amd.record import DecodeError
from golden.reference_vectors import (DERIVED_RFC5424, GELF, LTSV, RFC3164, RFC3164_CONFIG, RFC3164_VECTORS, RFC3164_YEAR, RFC5424,
                                      VECTORS)
from gpu_util import assert_same, device_path, host_path_blob
from test_oracle_golden import check_vector

pytestmark = pytest.mark.gpu

DECODERS = {RFC5424: RFC5424Decoder, LTSV: LTSVDecoder, GELF: GelfDecoder, RFC3164: RFC3164Decoder}
IDEO = "\u3000"   # IDEOGRAPHIC SPACE (White_Space, 3 bytes)
NBSP = "\u00a0"   # NO-BREAK SPACE (White_Space, 2 bytes)
ZWSP = "\u200b"   # ZERO WIDTH SPACE (NOT White_Space)
BOM = "\ufeff"


@pytest.fixture(scope="module")
def rfc():
    return RFC5424Decoder()


def both_paths(dec, oracle, lines, config=None):
    data, offsets = pack_lines(lines)
    oblob, ooffs = oracle.decode_batch(dec.fmt, data, offsets, config)
    (blob, offs), _ = host_path_blob(dec, data, offsets)
    assert_same(blob, offs, oblob, ooffs, lines)
    tables, _, _ = device_path(dec, data, offsets)
    blob2, offs2 = tables.to_host().serialize(dec.fmt, data, offsets, cfg=dec._cfg)
    assert_same(blob2, offs2, oblob, ooffs, lines)


@pytest.mark.parametrize("v", VECTORS + RFC3164_VECTORS, ids=[v["src"].split()[-1] for v in VECTORS + RFC3164_VECTORS])
def test_reference_vectors(v):
    """The reference's own decoder tests, run through Decoder.decode() on the GPU."""
    dec = DECODERS[v["fmt"]](v["config"])
    try:
        res = dec.decode(v["line"])
    except DecodeError as e:
        res = e
    check_vector(res, v)


def test_rfc5424_error_table(rfc, oracle):
    for line, err in DERIVED_RFC5424:
        with pytest.raises(DecodeError) as ei:
            rfc.decode(line)
        assert str(ei.value) == err, line
    both_paths(rfc, oracle, [ln for ln, _ in DERIVED_RFC5424])


def test_rfc5424_edge_semantics(rfc, oracle):
    hdr = "<13>1 2015-08-05T15:53:45Z h a p m "
    lines = [
        hdr + "-", hdr + "-x", hdr + f"- \t hello  {IDEO} ", "<13>1 2015-08-05T15:53:45Z  - -  - msg",
        hdr + '[a b="c"c="d"] m', hdr + '[a "b="c"] m', hdr + "[id ] m", hdr + r'[a b="x\\" c="\]\q\""] m',
        f"{BOM}<165>1 2003-10-11T22:14:15.003Z h a p m - {BOM}BOMmsg", "<+007>1 2015-08-05T15:53:45Z h a p m -",
        "<255>1 2015-08-05T15:53:45Z h a p m -", hdr + f"-    {NBSP}       {IDEO}x{IDEO}{NBSP}  ",
        hdr + f"- {ZWSP} zero width space is not White_Space {ZWSP}", hdr + "- ", hdr + "- café ",
        hdr + '[a b="é\\"]中"][c d="e"]  trailing  ', "<0>1 0000-01-01T00:00:00Z h a p m -",
        "<0>1 9999-12-31T23:59:59.999999999+25:59 h a p m -", "<0>1 1969-12-31T23:59:59.5-00:30 h a p m -",
        "<0>1 2016-12-31T23:59:60Z h a p m -", "<0>1 2016-12-31T23:59:60+01:00 h a p m -",
        "<0>1 2016-12-31T22:59:60-01:00 h a p m -", "<0>1 2015-08-05t15:53:45.1234567891234z h a p m -",
        "<0>1 2015-08-05T15:53:45.123+26:00 h a p m -", "<0>1 2015-08-05T15:53:45.123+05:60 h a p m -",
        "<0>1 9999-12-31T23:59:59.999999999-25:59 h a p m -", "<0>1 0000-01-01T00:00:00.000000001+25:59 h a p m -",
    ]
    both_paths(rfc, oracle, lines)
    r = rfc.decode(hdr + f"- \t hello  {IDEO} ")
    assert r.msg == "hello" and r.full_msg == hdr + "- \t hello"


def test_timestamp_bits_sweep(rfc, oracle):
    """f64 timestamps are bit-exact over the whole RFC3339 range, every fraction length and
    offsets (i128 -> f64 rounding + IEEE division on the GPU vs the CPU)."""
    rng = np.random.default_rng(3339)
    lines = []
    for i in range(20000):
        y = int(rng.integers(0, 10000)) if i % 3 else int(rng.integers(1990, 2040))
        mo, d = int(rng.integers(1, 13)), int(rng.integers(1, 29))
        h, mi, s = int(rng.integers(0, 24)), int(rng.integers(0, 60)), int(rng.integers(0, 60))
        nd = int(rng.integers(0, 13))
        frac = "." + "".join(str(int(x)) for x in rng.integers(0, 10, nd)) if nd else ""
        tz = "Z" if i % 2 else f"{'+-'[i % 4 // 2]}{int(rng.integers(0, 26)):02d}:{int(rng.integers(0, 60)):02d}"
        lines.append(f"<1>1 {y:04d}-{mo:02d}-{d:02d}T{h:02d}:{mi:02d}:{s:02d}{frac}{tz} h a p m -")
    both_paths(rfc, oracle, lines)


def test_empty_and_ragged_batches(rfc, oracle):
    both_paths(rfc, oracle, [])
    both_paths(rfc, oracle, [""])
    both_paths(rfc, oracle, ["", "<13>1 2015-08-05T15:53:45Z h a p m - x", "", ""])
    ok = "<13>1 2015-08-05T15:53:45Z h a p m - x"
    for n in (1, 63, 64, 65, 127, 128, 129, 1000):
        both_paths(rfc, oracle, [ok + "y" * (i % 7) for i in range(n)])


def test_long_lines_leave_the_lds_tile(rfc, oracle):
    """Lines far larger than a wave's LDS tile take the global-memory path of the same parser."""
    hdr = "<13>1 2015-08-05T15:53:45.5+02:00 host app 1 2 "
    big_sd = "[big " + " ".join(f'k{i}="{"v" * 50}"' for i in range(2000)) + "]"
    lines = [hdr + "- " + "m" * 100000, hdr + big_sd + " tail  ", hdr + "- short"] + \
            [hdr + "- " + "x" * (37 * i) for i in range(200)] + [hdr + big_sd[:-1]]
    both_paths(rfc, oracle, lines)


def test_cfg2_corpus_matches_oracle(rfc, oracle):
    """BASELINE config 2 tile: RFC5424 without structured data, ~256 B lines, 1 % invalid lines."""
    lines = synth.rfc5424_lines(200_000, cfg=2)
    data, offsets = synth.pack(lines)
    oblob, ooffs = oracle.decode_batch(RFC5424, data, offsets)
    (blob, offs), tab = host_path_blob(rfc, data, offsets)
    assert_same(blob, offs, oblob, ooffs, lines)
    st = tab.status
    assert (st != 0).sum() == 2000 and set(np.unique(st).tolist()) == set(range(0, 18))


def test_cfg4_sd_corpus_matches_oracle(rfc, oracle):
    """BASELINE config 4 tile: structured data (~12 pairs, escapes), ~512 B lines."""
    lines = synth.rfc5424_lines(100_000, cfg=4, sd=True)
    data, offsets = synth.pack(lines)
    oblob, ooffs = oracle.decode_batch(RFC5424, data, offsets)
    tables, _, _ = device_path(rfc, data, offsets)
    blob, offs = tables.to_host().serialize(RFC5424, data, offsets)
    assert_same(blob, offs, oblob, ooffs, lines)


def test_long_tail_lengths_match_oracle(rfc, oracle):
    """BASELINE config 5 shape: log-uniform 64 B .. 8 KiB line lengths (divergence stress)."""
    lines = synth.rfc5424_lines(30_000, cfg=5, sd=True, long_tail=True)
    both_paths(rfc, oracle, lines)


@pytest.mark.parametrize("knobs", [
    {"lines_per_group": 64}, {"lines_per_group": 32}, {"lines_per_group": 8},
    {"lines_per_group": 1}, {"lines_per_group": 48}, {"waves_per_cu": 1}, {"tile_cap": 4096},
    {"tile_cap": 40960, "lines_per_group": 64}, {"chunk_lines": 64}, {"chunk_lines": 5000, "tile_cap": 8192},
    {"force_head": 1}, {"force_head": 1, "tile_cap": 4096}, {"force_head": 1, "tile_cap": 40960, "lines_per_group": 7}, {"no_head": 1},
], ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()))
def test_kernel_variants_are_bit_identical(oracle, knobs):
    """Every launch shape of the RFC5424 kernel (64..1 lines
    per group, one wave per CU, tiles smaller than a group so that lines fall back to the global
    reader, tiles larger than the register window so that the tail loop runs) gives the oracle's
    bytes on all three corpora shapes.  The shapes are set through fg_set_launch_opts: the library reads no environment."""
    rfc = RFC5424Decoder()
    rfc.set_launch_opts(**knobs)
    lines = (synth.rfc5424_lines(20_000, cfg=2) + synth.rfc5424_lines(6_000, cfg=4, sd=True) +
             synth.rfc5424_lines(3_000, cfg=5, sd=True, long_tail=True))
    data, offsets = synth.pack(lines)
    oblob, ooffs = oracle.decode_batch(RFC5424, data, offsets)
    tables, _, _ = device_path(rfc, data, offsets)
    blob, offs = tables.to_host().serialize(RFC5424, data, offsets)
    assert_same(blob, offs, oblob, ooffs, lines)


def test_entry_table_overflow_is_reported(rfc):
    import torch

    lines = synth.rfc5424_lines(2000, cfg=4, sd=True)
    data, offsets = synth.pack(lines)
    tables, _, _ = device_path(rfc, data, offsets, ent_cap=100)
    used = int(tables.column("ent_used").view(torch.int64)[0].item())
    assert used > 100
    meta = tables.column("meta").view(torch.int32).cpu().numpy()
    assert ((meta & 0xFF) == 0xFE).sum() > 0


@pytest.mark.parametrize("fmt", [RFC5424, LTSV, GELF])
@pytest.mark.parametrize("share", [0.15, 0.5, 0.9])
def test_rows_that_are_not_flagged_stay_valid_when_the_entry_table_overflows(oracle, fmt, share):
    """include/fg_hip.h: "tables are valid except status == FG_ST_OVERFLOW rows".  A wave whose request straddles the end of the
    table keeps the slots of the lines before the cut; they must stay theirs (the partial allocation was not
    committed, a later group of the same wave was handed the same slots)."""
    if fmt == RFC5424:
        dec, lines, cfg = RFC5424Decoder(), synth.rfc5424_lines(6000, cfg=4, sd=True), None
    elif fmt == LTSV:
        dec, lines, cfg = LTSVDecoder(synth.LTSV_CONFIG), synth.ltsv_lines(6000), synth.LTSV_CONFIG
    else:
        dec, lines, cfg = GelfDecoder(), synth.gelf_lines(6000), None
    data, offsets = synth.pack(lines)
    oblob, ooffs = oracle.decode_batch(fmt, data, offsets, cfg)
    full, _, _ = device_path(dec, data, offsets)
    need = int(full.to_host().a["ent_count"].sum())
    cap = max(64, int(need * share))
    tables, _, _ = device_path(dec, data, offsets, ent_cap=cap)
    host = tables.to_host(allow_overflow=True)
    st = host.status
    over = st == 0xFE
    assert over.any() and not over.all()
    ok_rows = np.flatnonzero(~over)
    first, count = host.a["ent_first"][ok_rows].astype(np.int64), host.a["ent_count"][ok_rows].astype(np.int64)
    assert bool(((first + count <= cap) | (count == 0)).all()), "a row that is not flagged owns slots beyond ent_cap"
    # slices of unflagged rows are disjoint
    owner = np.zeros(cap + 1, np.int32)
    np.add.at(owner, first[count > 0], 1)
    np.add.at(owner, (first + count)[count > 0], -1)
    assert int(np.cumsum(owner).max()) <= 1, "two rows own the same entry slots"
    blob, offs = host.serialize(fmt, data, offsets, cfg=dec._cfg)
    bad = [int(i) for i in ok_rows
           if blob[int(offs[i]):int(offs[i + 1])].tobytes() != oblob[int(ooffs[i]):int(ooffs[i + 1])].tobytes()]
    assert not bad, f"{len(bad)} unflagged rows differ from the oracle, first: line {bad[0]} {lines[bad[0]]!r}"


def test_fuzz_mutations_match_oracle(rfc, oracle):
    """Random byte mutations of valid lines (ASCII + a few multi-byte scalars): results, including
    WHICH error is reported first, must match the oracle; nothing may crash."""
    rng = np.random.default_rng(5424)
    base = synth.rfc5424_lines(3000, cfg=4, sd=True, invalid_frac=0) + synth.rfc5424_lines(3000, cfg=2, invalid_frac=0)
    alphabet = [b" ", b"[", b"]", b'"', b"\\", b"=", b"<", b">", b"-", b"1", b"Z", b":", b".", b"+", b"\t", b"",
                "é".encode(), IDEO.encode(), NBSP.encode(), b"\x7f", b"T"]
    lines = []
    for ln in base:
        b = bytearray(ln)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, min(len(b), 120)))
            if b[pos] >= 0x80:
                continue
            b[pos:pos + 1] = alphabet[int(rng.integers(0, len(alphabet)))]
        lines.append(bytes(b))
    both_paths(rfc, oracle, lines)


def test_full_size_replicas_are_identical(rfc, oracle):
    """Size-independent property at scale: spans are line-relative, so decoding R back-to-back
    replicas of a tile yields R identical copies of the tile's table rows; replica 0 == oracle."""
    lines = synth.rfc5424_lines(250_000, cfg=2)
    data, offsets = synth.pack(lines)
    reps = 40  # 10 M lines, ~2.5 GB resident
    tables, d_bytes, d_offsets = device_path(rfc, data, offsets, ent_cap=1024, reps=reps)
    n = len(lines)
    for name in ("meta", "ts", "hostname", "appname", "procid", "msgid", "msg", "full_msg", "ent_count"):
        col = tables.column(name).view(reps, -1)
        assert bool((col == col[0:1]).all()), name
    oblob, ooffs = oracle.decode_batch(RFC5424, data, offsets)
    blob, offs = tables.to_host().serialize(RFC5424, data, offsets, 0, n)
    assert_same(blob, offs, oblob, ooffs, lines)


@pytest.mark.parametrize("shape", ["cfg3_gelf", "cfg4_sd", "cfg5_long_tail", "ltsv"])
def test_entry_workloads_at_scale_replicas_and_checksums(rfc, oracle, shape):
    """The entry-producing configurations at millions of lines: R back-to-back replicas of a tile
    decode to R identical fixed-column blocks and identical per-line entry counts, the entry table
    holds exactly R x the tile's entries (a checksum of checksums over names / values / types that
    does not depend on slot placement), and replica 0 materialises to the oracle's bytes."""
    import torch

    if shape == "cfg3_gelf":
        dec, fmt, cfg, lines = GelfDecoder(), GELF, None, synth.gelf_lines(120_000)
    elif shape == "ltsv":
        dec, fmt, cfg, lines = LTSVDecoder(synth.LTSV_CONFIG), LTSV, synth.LTSV_CONFIG, synth.ltsv_lines(120_000)
    elif shape == "cfg4_sd":
        dec, fmt, cfg, lines = rfc, RFC5424, None, synth.rfc5424_lines(120_000, cfg=4, sd=True)
    else:
        dec, fmt, cfg, lines = rfc, RFC5424, None, synth.rfc5424_lines(40_000, cfg=5, sd=True, long_tail=True)
    data, offsets = synth.pack(lines)
    reps = 16
    tables, _, _ = device_path(dec, data, offsets, reps=reps)
    n = len(lines)
    for name in ("meta", "ts", "hostname", "appname", "procid", "msgid", "msg", "full_msg", "ent_count"):
        col = tables.column(name).view(reps, -1)
        assert bool((col == col[0:1]).all()), name
    # entries: placement differs between replicas, content must not
    used = int(tables.column("ent_used").view(torch.int64)[0].item())
    cnt = tables.column("ent_count").view(torch.int32).view(reps, n)
    per_rep = int(cnt[0].sum().item())
    # (slots are reserved in per-wave chunks of at most 4096, sized so that all waves together strand at most 1/16 of the table
    #  -- include/fg_hip.h, ent_used: `used` counts the reserved slots -- the entries plus what every wave left of its last chunk)
    assert per_rep > 0 and per_rep * reps <= used <= per_rep * reps + tables.ent_cap // 16 + 2_500_000
    first = tables.column("ent_first").view(torch.int32).view(reps, n).to(torch.int64)
    name = tables.column("ent_name").view(torch.int64)[:used]
    val = tables.column("ent_val").view(torch.int64)[:used]
    typ = tables.column("ent_type")[:used].to(torch.int64)
    flg = tables.column("ent_flags")[:used].to(torch.int64)
    h = (name * 0x9E3779B1 + val * 0x85EBCA77 + typ * 0xC2B2AE3D + flg * 0x27D4EB2F)  # wraps mod 2^64
    csum = torch.cumsum(torch.cat([torch.zeros(1, dtype=torch.int64, device=h.device), h]), 0)
    line_sum = csum[first + cnt.to(torch.int64)] - csum[first]   # per line, per replica
    assert bool((line_sum == line_sum[0:1]).all()), "entry content differs between replicas"
    oblob, ooffs = oracle.decode_batch(fmt, data, offsets, cfg)
    blob, offs = tables.to_host().serialize(fmt, data, offsets, 0, n, cfg=dec._cfg)
    assert_same(blob, offs, oblob, ooffs, lines)


def test_host_path_slices_and_entries(rfc, oracle):
    """fg_decode_batch on a batch large enough to be cut into several ~32 MiB slices on two streams
    (rows land at their final position, entries share one counter): same bytes as the oracle."""
    lines = synth.rfc5424_lines(160_000, cfg=4, sd=True) + synth.rfc5424_lines(150_000, cfg=2)
    data, offsets = synth.pack(lines)
    assert data.size > 3 * (32 << 20)
    oblob, ooffs = oracle.decode_batch(RFC5424, data, offsets)
    (blob, offs), tab = host_path_blob(rfc, data, offsets)
    assert_same(blob, offs, oblob, ooffs, lines)
    total = int(tab.a["ent_count"].sum())
    assert total <= tab.ent_used <= total * 1.02 + 2_500_000  # (reserved in per-wave chunks)


# ------------------------------------------------------------------------------------- LTSV
def test_ltsv_corpus_matches_oracle(oracle):
    dec = LTSVDecoder(synth.LTSV_CONFIG)
    lines = synth.ltsv_lines(60_000)
    data, offsets = synth.pack(lines)
    oblob, ooffs = oracle.decode_batch(LTSV, data, offsets, synth.LTSV_CONFIG)
    (blob, offs), tab = host_path_blob(dec, data, offsets)
    assert_same(blob, offs, oblob, ooffs, lines)
    assert set(np.unique(tab.status).tolist()) == set(range(0, 10))
    both_paths(dec, oracle, synth.ltsv_lines(5000, long_tail=True), synth.LTSV_CONFIG)


def test_ltsv_semantics(oracle):
    cfg = {"input": {"ltsv_schema": {"n": "u64", "b": "bool", "f": "f64", "i": "i64", "s": "string", "x_f64": "F64"},
                     "ltsv_suffixes": {"f64": "_f64", "bool": "", "i64": "_i"}}}
    dec = LTSVDecoder(cfg)
    lines = [
        "time:1\thost:h\tnovalue\t\t_x:y\tn:7", "host:h", "time:1", "time:x\thost:h", "time:1\thost:h\tlevel:8",
        "time:1\thost:h\tlevel:x", "time:1\thost:h\tb:True", "time:1\thost:h\tf:x", "time:1\thost:h\ti:1.0",
        "time:1\thost:h\tn:-1", "level:9\ttime:x", "time:[1]\thost:h", "time:nan\thost:h", "time:-inf\thost:h",
        "time:1\ttime:2\thost:a\thost:b", "", "\t\t", ":", "time:1\thost:", "time:[]\thost:h", "time:[\thost:h",
        "time:1\thost:h\tf:1e400\tf:-0\tf:.5\tf:5.\tf:+1E-3\tf:0x1", "time:1\thost:h\tx_f64:2.5\tf:2.5\ti:-9223372036854775808",
        "time:1\thost:h\ti:9223372036854775808", "time:1\thost:h\tn:18446744073709551615\tn:18446744073709551616",
        "time:1\thost:h\tb:false\tb:true\ts:str\tunknown:v\ta:b:c\tlevel:+007",
        "time:[2015-08-05T15:53:45.637824+01:30]\thost:h", "time:[10/Oct/2000:13:55:36 -0700]\thost:h",
        "time:10/Oct/2000:13:55:36.123456789123 +0530\thost:h", "time:[31/Feb/2000:13:55:36 -0700]\thost:h",
        "time:[10/oct/2000:13:55:36 -0700]\thost:h", "time:[10/Oct/2000:13:55:60 -0700]\thost:h",
        "time:[1/Jan/+2000:00:00:00 +0000]\thost:h", "time:[01/Jan/-0001:00:00:00 -0030]\thost:h",
        "time:1438790025.99\thost:h", "time:1.7976931348623159e308\thost:h", "time:9007199254740993\thost:h",
        "time:0.1000000000000000055511151231257827021181583404541015625\thost:h",
        "time:123456789012345678901234567890e-10\thost:h",
        "time:2.47032822920623272088284396434110686182529901307162382212792841250337753635104375932649918180817996189"
        "898282347722858865463328355177969898199387398005390939063150356595155702263922908583924491051844359318028499"
        "365361525003193704576782492193656236698636584807570015857692699037063119282795585513329278343384093519780155"
        "312465972635795746227664652728272200563740064854999770965994704540208281662262378573934507363390079677619305"
        "775067401763246736009689513405355374585166611342237666786041621596804619144672918403005300575308490487653917"
        "113865916462395249126236538818796362393732804238910186723484976682350898633885879256283027559956575244555072"
        "551893136908362547791869486679949683240497058210285131854513962138377228261454376934125320985913276672363281"
        "25e-324\thost:h",
    ]
    both_paths(dec, oracle, lines, cfg)
    r = dec.decode("time:1\thost:h\tx_f64:2.5\tf:2.5\tb:true\ti:3")
    assert [(k, v.kind) for k, v in r.sd[0].pairs] == [("_x_f64", "F64"), ("_f_f64", "F64"), ("_b", "Bool"), ("_i_i", "I64")]


def test_ltsv_f64_rounding_fuzz(oracle):
    """Typed f64 values through all three dec2flt stages on the GPU (incl. forced Decimal path)."""
    from decimal import Decimal, getcontext
    from fractions import Fraction
    import struct

    getcontext().prec = 1200
    rng = np.random.default_rng(17)
    cfg = {"input": {"ltsv_schema": {"f": "f64"}}}
    dec = LTSVDecoder(cfg)
    lines = []
    for i in range(4000):
        e = int(rng.integers(1, 2046)) if i % 10 else 0
        m = int(rng.integers(0, 1 << 52))
        x = struct.unpack("<d", struct.pack("<Q", (e << 52) | m))[0]
        y = struct.unpack("<d", struct.pack("<Q", ((e << 52) | m) + 1))[0]
        if y == float("inf"):
            continue
        mid = (Fraction(x) + Fraction(y)) / 2
        d = Decimal(mid.numerator) / Decimal(mid.denominator)
        s = format(d, "e")
        nd = int(rng.integers(1, 40))
        rnd = "".join(str(int(v)) for v in rng.integers(0, 10, nd)) + "e" + str(int(rng.integers(-340, 310)))
        lines.append(f"time:{s}\thost:h\tf:{s.replace('e', '1e')}\tf:{rnd}\tf:{repr(x)}")
    both_paths(dec, oracle, lines, cfg)


# ------------------------------------------------------------------------------------- GELF
def test_gelf_corpus_matches_oracle(oracle):
    dec = GelfDecoder()
    lines = synth.gelf_lines(60_000)
    data, offsets = synth.pack(lines)
    oblob, ooffs = oracle.decode_batch(GELF, data, offsets)
    (blob, offs), tab = host_path_blob(dec, data, offsets)
    assert_same(blob, offs, oblob, ooffs, lines)
    assert len(set(np.unique(tab.status).tolist())) >= 10


def test_gelf_semantics(oracle):
    dec = GelfDecoder()
    many = "{" + ",".join(f'"k{(i * 7919) % 100:03d}":{i}' for i in range(100)) + ',"host":"h"}'
    deep_ok = '{"host":"h","a":' + "[" * 510 + "]" * 510 + "}"
    deep_bad = '{"host":"h","a":' + "[" * 512 + "]" * 512 + "}"
    lines = [
        '{"host":"h","b":1,"a":2,"_c":null,"B":true,"a":-3}', "[1,2]", '{"host":"h"} x', '{"a":1}', '{"host":1}',
        '{"host":"h","level":-1}', '{"host":"h","level":1.0}', '{"host":"h","version":1}', '{"host":"h","short_message":1}',
        '{"host":"h","full_message":null}', '{"host":"h","x":{"y":1}}', '{"host":"h\\u00e9\\n\\ud83d\\ude00","timestamp":1}',
        '{"host":"a\nb","timestamp":1}', '{"host":"a\tb"}', '{"host":1,"_x":[]}', "", " ", "null", "true", '"str"', "12", "-",
        "{}", "{ }", '{"host":"h",}', '{,"host":"h"}', '{"host" "h"}', '{"host":"h"', '{"host":"h"}}', "[", "[1,]", "[,1]",
        '{"host":"h","a":[1,{"b":[true,false,null,"x",-1.5e3]}],"c":1}', '{"host":"h","a":[1 2]}', '{"host":"h","a":{"b":1,}}',
        '{"host":"h","a":tru}', '{"host":"h","a":nul}', '{"host":"h","a":falsE}', '{"host":"h","a":01}', '{"host":"h","a":1.}',
        '{"host":"h","a":-}', '{"host":"h","a":1e}', '{"host":"h","a":1e999}', '{"host":"h","a":-1e999}', '{"host":"h","a":1e-999}',
        '{"host":"h","a":18446744073709551615,"b":18446744073709551616,"c":-9223372036854775808,"d":-9223372036854775809}',
        '{"host":"h","a":0.1,"b":1385053862.3072,"c":123456789012345678901234567890.5e-5,"d":-0,"e":-0.0,"f":1E+2}',
        '{"host":"h","timestamp":-5}', '{"host":"h","timestamp":18446744073709551615}', '{"host":"h","timestamp":true}',
        '{"h\\u006fst":"escaped key","_a\\"b":"q","_a\\u0022b":"dup wins","a\\\\b":1,"a\\/b":2,"a\\tb":3}',
        '{"host":"h","s":"\\ud83d"}', '{"host":"h","s":"\\ud83d\\u0041"}', '{"host":"h","s":"\\udc00"}', '{"host":"h","s":"\\u12g4"}',
        '{"host":"h","s":"\\x"}', '{"host":"h","s":"abc', '{"host":"h","s":"\\', '{"host":"h","s":"\\u00"}', '{"host":"h","s":"\\u0000"}',
        '{"host":"h","version":"1.0"}', '{"host":"h","version":"1.\\u0031"}', '{"host":"h","version":"1.2"}', '{"host":"h","level":7}',
        '{"host":"h","level":8}', '{"host":"h","level":"1"}', '{"_":1,"__":2,"":3,"host":"h"}',
        '{"host":"line1\nline2\nline3","short_message":"with\nnewline","_k\ney":"v"}',  # retry accepted
        '{"host":"a\nb",\n"x":1}',  # retry turns the structural newline into a syntax error
        '{"host":"a\\\nb"}',  # backslash + raw LF: escaped backslash then n (retry)
        '{"host":"a\nb\\\nc","k\\\n":"v\\\n"}', '{"host":"a\nb","s":"\\ud83d\n\\ude00"}', '{"host":"a\rb"}', '{"host":"a\x00b"}',
        '\n{"host":"h"}\n', '{"host":"h","a":"x\x7fy"}', '{"host":"h","a":"caf\u00e9 \u4e2d\u6587 \U0001F600"}',
        many, deep_ok, deep_bad, '{"host":"h","big":' + "9" * 400 + "}", '{"host":"h","big":0.' + "9" * 400 + "}",
        '{"host":"h","e":1e2147483648}', '{"host":"h","e":0e2147483648}', '{"host":"h","e":1e-2147483649}',
        '{"zz":[],"host":5}', '{"timestamp":"x","level":99,"host":"h"}', '{"version":"9","timestamp":"x","host":"h"}',
    ]
    both_paths(dec, oracle, lines)
    r = dec.decode('{"host":"h","b":1,"a":2,"_c":null,"B":true,"a":-3}')
    assert [(k, v.kind, v.value) for k, v in r.sd[0].pairs] == [
        ("_B", "Bool", True), ("_c", "Null", None), ("_a", "I64", -3), ("_b", "U64", 1)]


def test_gelf_fuzz_mutations_match_oracle(oracle):
    dec = GelfDecoder()
    rng = np.random.default_rng(808)
    base = synth.gelf_lines(6000, invalid_frac=0)
    alphabet = [b'"', b"\\", b",", b":", b"{", b"}", b"[", b"]", b" ", b"\n", b"\t", b"0", b"-", b".", b"e", b"u", b"n",
                b"true", b"null", b"", b"\\u00e9", b"\\n", "é".encode(), b"\x01", b"_"]
    lines = []
    for ln in base:
        b = bytearray(ln)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, len(b)))
            if b[pos] >= 0x80:
                continue
            b[pos:pos + 1] = alphabet[int(rng.integers(0, len(alphabet)))]
        lines.append(bytes(b))
    both_paths(dec, oracle, lines)


# -------------------------------------------------------------------------- sharding (N > 1 path)
def test_sharded_decode_equals_oracle(rfc, oracle):
    """Byte-balanced 8-way shard plan, each shard decoded on its own, ordered host gather."""
    from flowgger_amd import shard

    lines = synth.rfc5424_lines(40_000, cfg=4, sd=True)
    data, offsets = synth.pack(lines)
    tab = shard.decode_sharded(lambda d, o, k: rfc.decode_packed(np.ascontiguousarray(d), o), data, offsets, 8)
    blob, offs = tab.serialize(RFC5424, data, offsets)
    oblob, ooffs = oracle.decode_batch(RFC5424, data, offsets)
    assert_same(blob, offs, oblob, ooffs, lines)


def test_cfg5_mixed_formats_ordered_merge(rfc, oracle):
    from flowgger_amd import shard

    ltsv = LTSVDecoder(synth.LTSV_CONFIG)
    a = synth.rfc5424_lines(3000, cfg=5, sd=True, long_tail=True)
    b = synth.ltsv_lines(2000, long_tail=True)
    rng = np.random.default_rng(55)
    tag = rng.permutation(np.array([0] * len(a) + [1] * len(b)))
    parts = []
    for t, dec, lines in ((0, rfc, a), (1, ltsv, b)):
        d, o = synth.pack(lines)
        (blob, offs), _ = host_path_blob(dec, d, o)
        parts.append((np.nonzero(tag == t)[0], blob, offs))
    blob, offs = shard.ordered_merge(parts)
    it = {0: iter(a), 1: iter(b)}
    for i, t in enumerate(tag):
        want = oracle.decode(int(t), next(it[int(t)]), synth.LTSV_CONFIG if t else None)
        assert blob[int(offs[i]):int(offs[i + 1])].tobytes() == want, i


def test_cfg5mix_one_million_tagged_lines_match_the_oracle(oracle):
    """BASELINE configs[4] as specified: a 50/50 tagged RFC5424 + LTSV stream, log-uniform 64 B..8 KiB, decoded as
    two sub-batches on the GPU; each sub-batch's Records == the oracle's; fg_merge_tables puts the rows (and their entries)
    back at their arrival positions; fg_ordered_merge of the Records gives the arrival-order stream the reference's two
    inputs would have produced.  1 M lines, ~1.7 GB."""
    import os

    from flowgger_amd import shard

    n = 1_000_000
    tag, (la, ia), (lb, ib) = synth.mixed_cfg5(n)
    assert len(la) + len(lb) == n and abs(len(la) - len(lb)) < n // 50
    threads = min(len(os.sched_getaffinity(0)), 64)
    parts, recs, index = [], [], [ia, ib]
    for fmt, lines, ix, dec, cfg in ((RFC5424, la, ia, RFC5424Decoder(), None),
                                     (LTSV, lb, ib, LTSVDecoder(synth.LTSV_CONFIG), synth.LTSV_CONFIG)):
        data, offsets = synth.pack(lines)
        tables, _, _ = device_path(dec, data, offsets, ent_cap=int(offsets[-1]) // 16 + (1 << 22))
        host = tables.to_host_pinned()
        blob, offs = host.serialize(fmt, data, offsets, cfg=dec._cfg)
        oblob, ooffs = oracle.decode_batch(fmt, data, offsets, cfg, threads=threads)
        assert_same(blob, offs, oblob, ooffs, lines)
        parts.append(host)
        recs.append((ix, oblob, ooffs))
        del data, tables
    merged, src = shard.merge_tables(parts, index)
    assert merged.n == n and np.array_equal(src, tag)
    ent0 = 0
    for k, (p, ix) in enumerate(zip(parts, index)):
        ix = ix.astype(np.int64)
        for col in ("meta", "ts", "ent_count"):
            assert np.array_equal(merged.a[col][ix], p.a[col][: p.n]), col
        for col in ("hostname", "appname", "procid", "msgid", "msg", "full_msg"):
            assert np.array_equal(merged.a[col].reshape(-1, 2)[ix], p.a[col].reshape(-1, 2)[: p.n]), col
        has = p.a["ent_count"][: p.n] != 0
        assert np.array_equal(merged.a["ent_first"][ix][has], p.a["ent_first"][: p.n][has] + np.uint32(ent0))
        u = p.ent_used
        for col, w in (("ent_name", 2), ("ent_val", 1), ("ent_type", 1), ("ent_flags", 1)):
            assert np.array_equal(merged.a[col][ent0 * w:(ent0 + u) * w], p.a[col][: u * w]), col
        ent0 += u
    # the arrival-order Record stream: record i is the oracle's record of the line that arrived i-th
    blob, offs = shard.ordered_merge(recs)
    sizes = np.diff(offs.astype(np.int64))
    for ix, ob, oo in recs:
        assert np.array_equal(sizes[ix.astype(np.int64)], np.diff(oo.astype(np.int64)))
    rng = np.random.default_rng(4)
    for i in rng.integers(0, n, 2000):
        k = int(tag[i])
        j = int(np.searchsorted(index[k], i))
        ix, ob, oo = recs[k]
        assert blob[int(offs[i]):int(offs[i + 1])].tobytes() == ob[int(oo[j]):int(oo[j + 1])].tobytes()


# ---------------------------------------------------------------------- C++ host mirror + framers
@pytest.mark.parametrize("fmt,framing", [("rfc5424", "line"), ("gelf", "nul"), ("ltsv", "syslen"), ("rfc5424", "gpu-line"),
                                         ("ltsv", "gpu-line"), ("gelf", "gpu-nul")])
def test_cpp_host_mirror_and_batching_splitters(tmp_path, oracle, fmt, framing):
    """fg::Decoder / fg::BatchingSplitter (C++ mirror of the trait and of the three splitters):
    Ok records in input order == oracle, error lines formatted like line_splitter.rs:37-39."""
    import subprocess
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "host_mirror_test"
    subprocess.run(["g++", "-std=c++17", "-O1", str(root / "tests/native/host_mirror_test.cpp"), "-o", str(exe),
                    f"-L{root / 'flowgger_amd'}", "-lfg_hip", f"-Wl,-rpath,{root / 'flowgger_amd'}",
                    "-L/opt/rocm/lib", "-lamdhip64"], check=True)
    code = {"rfc5424": RFC5424, "ltsv": LTSV, "gelf": GELF}[fmt]
    cfg = synth.LTSV_CONFIG if fmt == "ltsv" else None
    lines = {"rfc5424": lambda: synth.rfc5424_lines(3000, cfg=4, sd=True), "gelf": lambda: synth.gelf_lines(3000),
             "ltsv": lambda: synth.ltsv_lines(3000)}[fmt]()
    lines = [ln for ln in lines if b"\n" not in ln and b"\0" not in ln]
    lines[7] = lines[7] + b"  "
    raw = bytearray()
    expect_lines = []
    for i, ln in enumerate(lines):
        if framing in ("line", "gpu-line"):
            raw += ln + (b"\r\n" if i % 5 == 0 else b"\n")
        elif framing in ("nul", "gpu-nul"):
            raw += ln + b"\0"
        else:
            raw += str(len(ln)).encode() + b" " + ln
        expect_lines.append(ln)
    bad_utf8 = b"<13>1 2015-08-05T15:53:45Z h a p m - \xff\xfe"
    if framing != "syslen":  # the reference unwrap()-panics on invalid UTF-8 under syslen framing
        raw += bad_utf8 + (b"\n" if framing in ("line", "gpu-line") else b"\0")
    if framing.startswith("gpu-"):
        raw += lines[11]  # an unterminated last frame is a frame
        expect_lines.append(lines[11])
    f = tmp_path / "in.bin"
    f.write_bytes(bytes(raw))
    # batch size (lines) for the host framers, chunk size (bytes) for the GPU-framing splitter: small,
    # so that frames straddle chunks and the carry-over logic is exercised
    p = subprocess.run([str(exe), fmt, framing, str(f), "257" if not framing.startswith("gpu-") else "20011"], capture_output=True)
    assert p.returncode == 0, p.stderr[-2000:]
    got_ok = [bytes.fromhex(x) for x in p.stdout.decode().split()]
    got_err = p.stderr.decode("utf-8", "replace").splitlines()
    want_ok, want_err = [], []
    for ln in expect_lines:
        c = oracle.decode(code, ln, cfg)
        if c[0] == 0:
            want_ok.append(c)
        else:
            msg = c[5:].decode()
            t = ln.decode("utf-8", "replace").strip()
            if not (framing in ("nul", "gpu-nul") and t == ""):
                want_err.append(f"{msg}: [{t}]")
    if framing.startswith("gpu-"):
        # the invalid frame sits before the unterminated last line
        tail_errs = want_err[-1:] if oracle.decode(code, lines[11], cfg)[0] != 0 else []
        want_err = want_err[:len(want_err) - len(tail_errs)] + ["Invalid UTF-8 input"] + tail_errs
    elif framing != "syslen":
        want_err.append("Invalid UTF-8 input")
    else:
        want_err.append("Can't read message's length")
    assert got_ok == want_ok
    assert got_err == want_err


# ---------------------------------------------------------------------------------------------
# GPU framing + UTF-8 validation: BufRead::lines() / split(0) + str::from_utf8
# ---------------------------------------------------------------------------------------------
def _frames_reference(raw: bytes, framing: str):
    """The reference splitters' framing, restated with the standard library: list of
    (start, end_with_terminator, line_bytes_without_terminator, is_valid_utf8)."""
    delim = b"\n" if framing == "line" else b"\0"
    out, pos = [], 0
    while pos < len(raw):
        k = raw.find(delim, pos)
        end = len(raw) if k < 0 else k + 1
        body = raw[pos:end]
        if k >= 0:
            body = body[:-1]
            if framing == "line" and body.endswith(b"\r"):  # lines(): strips "\n", then one "\r"
                body = body[:-1]
        try:
            body.decode("utf-8")
            ok = True
        except UnicodeDecodeError:
            ok = False
        out.append((pos, end, body, ok))
        pos = end
    return out


def _utf8_torture():
    good = ["plain ascii", "caf\u00e9", "\u20ac euro", "\U0001F600 smile", "\u07ff\u0800\uffff\U00010000\U0010ffff", ""]
    bad = [b"\xff", b"\xc0\xaf", b"\xc1\xbf", b"\xe0\x80\x80", b"\xe0\x9f\xbf", b"\xed\xa0\x80", b"\xed\xbf\xbf", b"\xf0\x80\x80\x80",
           b"\xf0\x8f\xbf\xbf", b"\xf4\x90\x80\x80", b"\xf5\x80\x80\x80", b"\x80", b"\xbf", b"a\x80b", b"\xc2", b"\xe2\x82", b"\xf0\x9f\x98",
           b"\xc2\x41", b"\xe2\x28\xa1", b"\xe2\x82\x28", b"\xf0\x28\x8c\xbc", b"\xf0\x90\x28\xbc", b"\xf0\x28\x8c\x28", b"\xf8\x88\x80\x80\x80",
           b"ok\xe2\x82\xac then \xe2\x82 cut"]
    return [g.encode() for g in good] + bad


@pytest.mark.parametrize("framing", ["line", "nul"])
def test_gpu_framing_and_utf8_match_the_splitters(rfc, oracle, framing):
    import torch
    from flowgger_amd import _lib as L
    from flowgger_amd.tables import DeviceTables

    rng = np.random.default_rng(0x8f1)
    lines = synth.rfc5424_lines(40_000, cfg=2) + synth.rfc5424_lines(2_000, cfg=5, sd=True, long_tail=True)
    tort = _utf8_torture()
    delim = b"\n" if framing == "line" else b"\0"
    pieces = []
    for i, ln in enumerate(lines):
        if i % 7 == 3:
            ln = ln + b" " + tort[(i // 7) % len(tort)]
        if i % 11 == 5:
            ln = tort[(i // 11) % len(tort)] + b" " + ln  # the decoder then sees an unsupported BOM / garbage
        if framing == "line" and i % 5 == 1:
            ln = ln + b"\r"
        if i % 97 == 0:
            ln = b""  # empty frames are frames
        pieces.append(ln + delim)
    for tail in (b"", b"unterminated tail \xe2\x82", b"plain tail", b"\r"):
        raw = b"".join(pieces) + tail
        ref = oracle.frame(raw, framing)
        assert ref == _frames_reference(raw, framing)  # (the oracle's restatement and the standard library's verdicts agree)
        dev = torch.device("cuda", rfc.device)
        d_bytes = torch.cat([torch.frombuffer(bytearray(raw), dtype=torch.uint8), torch.zeros(32, dtype=torch.uint8)]).to(dev)[:len(raw) + 32]
        d_raw = d_bytes[:len(raw)]
        fr = L.FG_FRAME_LINE if framing == "line" else L.FG_FRAME_NUL
        # cap too small first: must report the need, not overflow
        d_offsets, d_bad, n = rfc.frame_device(d_raw, fr, cap_frames=8)
        assert n == len(ref)
        offs = d_offsets[:n + 1].cpu().numpy().astype(np.uint64)
        bad = d_bad[:n].cpu().numpy()
        assert np.array_equal(offs[:-1], np.array([r[0] for r in ref], np.uint64))
        assert int(offs[-1]) == len(raw)
        want_bad = np.array([0 if r[3] else 1 for r in ref], np.uint8)
        assert np.array_equal(bad, want_bad), f"first UTF-8 verdict mismatch at frame {int(np.flatnonzero(bad != want_bad)[0])}"
        # decode the frames in place (terminators stripped in-kernel) == oracle on the bare lines
        tables = DeviceTables(n, len(raw) // 8 + 1024, dev)
        rfc.decode_frames_device(d_raw, d_offsets, n, tables, fr, d_bad)
        torch.cuda.synchronize(dev)
        host = tables.to_host()
        st = host.status
        assert np.array_equal(st == L.FG_ST_BAD_UTF8, want_bad == 1)
        good = [r[2] for r in ref if r[3]]
        gdata, goffs = synth.pack(good)
        oblob, ooffs = oracle.decode_batch(RFC5424, gdata, goffs)
        idx = np.flatnonzero(want_bad == 0)
        # serialise the good rows one by one against the framed buffer (spans are line-relative)
        raw_np = np.frombuffer(raw, np.uint8)
        blob, boffs = host.serialize(RFC5424, raw_np, offs)
        for j, i in enumerate(idx[:: max(1, len(idx) // 4000)]):
            jj = int(np.searchsorted(idx, i))
            a = blob[int(boffs[i]):int(boffs[i + 1])].tobytes()
            b = oblob[int(ooffs[jj]):int(ooffs[jj + 1])].tobytes()
            assert a == b, (int(i), ref[int(i)][2][:80])


def test_gpu_framing_block_boundaries(rfc):
    """Streams whose length is 0, 1, a multiple of the 16 KiB scan block, and sequences that
    straddle chunk / row / block boundaries."""
    import torch
    from flowgger_amd import _lib as L

    dev = torch.device("cuda", rfc.device)
    cases = [b"\n", b"x", b"\xe2\x82\xac\n" * 5461 + b"ab", (b"a" * 16383 + b"\n"), (b"a" * 16382 + b"\xc3\xa9" * 8192 + b"\n"),
             b"a" * 16383 + b"\xc3", b"a" * 16381 + b"\xe2\x82\xac" + b"\xe2\x82\xac\n", b"a" * 1023 + b"\xf0\x9f\x98\x80" * 5000,
             b"\n" * 40000, b"a" * 32768]
    for raw in cases:
        ref = _frames_reference(raw, "line")
        d_bytes = torch.cat([torch.frombuffer(bytearray(raw), dtype=torch.uint8), torch.zeros(32, dtype=torch.uint8)]).to(dev)
        d_offsets, d_bad, n = rfc.frame_device(d_bytes[:len(raw)], L.FG_FRAME_LINE)
        assert n == len(ref), (raw[:20], n, len(ref))
        offs = d_offsets[:n + 1].cpu().numpy()
        assert [int(x) for x in offs[:-1]] == [r[0] for r in ref] and int(offs[-1]) == len(raw)
        assert d_bad[:n].cpu().numpy().tolist() == [0 if r[3] else 1 for r in ref], raw[:20]


def test_frame_decode_batch_chunked_stream(rfc, oracle):
    """fg_frame_decode_batch over a stream cut into arbitrary chunks: the carried-over tails and the
    per-chunk results concatenate to exactly the frames of the whole stream."""
    from flowgger_amd import _lib as L

    lines = synth.rfc5424_lines(5_000, cfg=2) + synth.rfc5424_lines(1_500, cfg=4, sd=True)
    tort = _utf8_torture()
    raw = b"".join((ln + (b" " + tort[i % len(tort)] if i % 9 == 4 else b"")) + (b"\r\n" if i % 4 == 0 else b"\n")
                   for i, ln in enumerate(lines)) + b"last line without terminator"
    ref = _frames_reference(raw, "line")
    rng = np.random.default_rng(3)
    got = []  # (status, canonical bytes)
    pos, carry = 0, b""
    while pos < len(raw) or carry:
        step = int(rng.integers(1, 200_000))
        chunk = carry + raw[pos:pos + step]
        pos += step
        final = pos >= len(raw)
        tab, offs, used = rfc.frame_decode_batch(chunk, L.FG_FRAME_LINE, final=final)
        if tab is not None:
            blob, boffs = tab.serialize(RFC5424, np.frombuffer(chunk + b"\0" * 16, np.uint8), offs)
            st = tab.status
            for i in range(len(offs) - 1):
                got.append((int(st[i]), blob[int(boffs[i]):int(boffs[i + 1])].tobytes()))
        carry = chunk[used:]
        if final:
            assert carry == b""
            break
    assert len(got) == len(ref)
    good = [r[2] for r in ref if r[3]]
    gdata, goffs = synth.pack(good)
    oblob, ooffs = oracle.decode_batch(RFC5424, gdata, goffs)
    j = 0
    for (st, blob), r in zip(got, ref):
        if not r[3]:
            assert st == L.FG_ST_BAD_UTF8
            continue
        assert blob == oblob[int(ooffs[j]):int(ooffs[j + 1])].tobytes(), r[2][:60]
        j += 1


# ---------------------------------------------------------------------------------------------
# GELF encoder from the tables: GelfEncoder::encode, gelf_encoder.rs:59-115
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("src,extra", [
    ("rfc5424", None), ("rfc5424_sd", {"secret-token": "secret"}),
    ("rfc5424_sd", {"_k1_b": "replaced by gelf_extra", "host": "forced", "Zeta": "capital sorts first", "_a": "x\"y\\z\n\x01"}),
    ("ltsv", {"zz": "last", "_counter_u64": "shadow"}), ("long_tail", None)])
def test_gelf_encoder_matches_the_reference_pipeline(rfc, oracle, src, extra):
    """decode on the GPU -> encode on the GPU == the oracle's decode -> GelfEncoder::encode, byte for byte,
    for every line (sorted keys, later inserts win, gelf_extra last, escapes, Grisu2 timestamps)."""
    import torch

    if src == "ltsv":
        dec, fmt, cfg, lines = LTSVDecoder(synth.LTSV_CONFIG), LTSV, synth.LTSV_CONFIG, synth.ltsv_lines(20_000)
    elif src == "rfc5424":
        dec, fmt, cfg, lines = rfc, RFC5424, None, synth.rfc5424_lines(30_000, cfg=2)
    elif src == "long_tail":
        dec, fmt, cfg, lines = rfc, RFC5424, None, synth.rfc5424_lines(6_000, cfg=5, sd=True, long_tail=True)
    else:
        dec, fmt, cfg = rfc, RFC5424, None
        lines = synth.rfc5424_lines(20_000, cfg=4, sd=True)
        hdr = b"<13>1 2015-08-05T15:53:45.637824Z host app 1234 ID7 "
        lines += [hdr + b'[a x="1" x="2" y="3"][b x="4" request_id="r1" request_ip="r2" request_i="r3"] dup keys and 7-byte prefix ties',
                  hdr + b'[big ' + b" ".join(b'k%02d="v%d"' % (99 - i, i) for i in range(40)) + b"] more than 32 pairs",
                  hdr + b'[e esc="a\\"b\\\\c\\]d\\qe" ctl="tab\there"] escapes',
                  b"<13>1 2015-08-05T15:53:45Z - - - - - ", b"<191>1 1999-12-31T23:59:59.999999999+14:00  a  p m - no host"]
    data, offsets = synth.pack(lines)
    tables, d_bytes, d_offsets = device_path(dec, data, offsets)
    d_out, d_off = dec.encode_gelf_device(d_bytes, d_offsets, len(lines), tables, extra)
    torch.cuda.synchronize()
    out, off = d_out.cpu().numpy(), d_off.cpu().numpy()
    oblob, ooffs = oracle.decode_encode_gelf_batch(fmt, data, offsets, cfg, extra)
    assert int(off[-1]) == int(ooffs[-1]), (int(off[-1]), int(ooffs[-1]))
    bad = np.flatnonzero(off.astype(np.uint64) != ooffs)
    assert len(bad) == 0, (int(bad[0]), lines[int(bad[0]) - 1][:120])
    if not np.array_equal(out, oblob):
        i = int(np.searchsorted(ooffs, np.flatnonzero(out != oblob)[0], side="right") - 1)
        a, b = out[int(off[i]):int(off[i + 1])].tobytes(), oblob[int(ooffs[i]):int(ooffs[i + 1])].tobytes()
        raise AssertionError(f"line {i}: {lines[i][:100]!r}\n  gpu    {a!r}\n  oracle {b!r}")
    assert (np.diff(ooffs.astype(np.int64)) > 0).sum() > 0.9 * len(lines)


# ---------------------------------------------------------------------------------------------
# Every encoder x merger from the tables, records from all three decoders
# ---------------------------------------------------------------------------------------------
def _encode_corpus(src, rfc):
    hdr = b"<13>1 2015-08-05T15:53:45.637824Z host app 1234 ID7 "
    if src == "ltsv":
        return LTSVDecoder(synth.LTSV_CONFIG), LTSV, synth.LTSV_CONFIG, synth.ltsv_lines(6_000)
    if src == "gelf":
        lines = synth.gelf_lines(6_000)
        lines += [b'{"host":"h","short_message":"no timestamp: wall clock"}',
                  b'{"host":"h\\u00e9\\ud834\\udd1e","short_message":"a\\"b\\\\c\\/d\\n\\t\\u0001","full_message":"x\\u20acy","timestamp":1.5,"level":3,'
                  b'"_k\\u005f1":"v\\u0041","\\u005fesc":true,"plain":null,"_n":-5,"u":18446744073709551615,"f":1e21,"g":1.0e-7}',
                  b'{"host":"retry","short_message":"raw\nnewline and \\\nbackslash-newline","_k":"v\n"}',
                  b'{"host":"h","timestamp":253402300800,"short_message":"year 10000"}',
                  b'{"host":"h","timestamp":-62167219201,"short_message":"year -1"}',
                  b'{"host":"h","timestamp":1e300,"short_message":"saturating"}',
                  b'{"host":"h","_dup":"1","dup":"2","_dup":"3","sd_id":"x","_sd_id":"y","__u":"z"}']
        return GelfDecoder(), GELF, None, lines
    if src == "long_tail":
        return rfc, RFC5424, None, synth.rfc5424_lines(3_000, cfg=5, sd=True, long_tail=True)
    lines = synth.rfc5424_lines(6_000, cfg=2) + synth.rfc5424_lines(6_000, cfg=4, sd=True)
    lines += [hdr + b'[a x="1" x="2" y="3"][b x="4" request_id="r1" request_ip="r2" request_i="r3"] dup keys and 7-byte prefix ties',
              hdr + b'[big ' + b" ".join(b'k%02d="v%d"' % (99 - i, i) for i in range(40)) + b"] more than 32 pairs",
              hdr + b'[e esc="a\\"b\\\\c\\]d\\qe" ctl="tab\there" k:c="v:w"] escapes',
              hdr + b"[empty ] element without pairs", hdr + b"[] []  - ",
              b"<13>1 2015-08-05T15:53:45Z - - - - - ", b"<191>1 1999-12-31T23:59:59.999999999+14:00  a  p m - no host",
              b"<13>1 9999-12-31T23:59:59.9996Z h a p m - last millisecond", b"<13>1 0000-01-01T00:00:00+01:00 h a p m - year -1 in UTC",
              b"<13>1 9999-12-31T23:59:59+00:00 h a p m - x", b"<13>1 9999-12-31T23:59:59-23:00 h a p m - year 10000 in UTC",
              b"<255>1 1970-01-01T00:00:00.5Z h a p m - half", b"<13>1 1969-12-31T23:59:59.25Z h a p m - negative ts"]
    return rfc, RFC5424, None, lines


@pytest.mark.parametrize("src", ["rfc5424", "ltsv", "gelf", "long_tail"])
@pytest.mark.parametrize("enc", ["gelf", "ltsv", "rfc5424", "rfc3164", "passthrough"])
def test_encoders_and_mergers_match_the_reference_pipeline(rfc, oracle, src, enc):
    """decode on the GPU -> encode + frame on the GPU == the oracle's decode -> Encoder::encode -> Merger::frame, byte for
    byte, for every line and every merger; lines whose decode or encode fails produce nothing and the right status."""
    import torch

    import oracle_binding as OB
    from flowgger_amd import GelfEncoder, LTSVEncoder, PassthroughEncoder, RFC3164Encoder, RFC5424Encoder

    dec, fmt, cfg, lines = _encode_corpus(src, rfc)
    data, offsets = synth.pack(lines)
    tables, d_bytes, d_offsets = device_path(dec, data, offsets)
    cls, oenc, extra_key = {"gelf": (GelfEncoder, OB.ENC_GELF, "gelf_extra"), "ltsv": (LTSVEncoder, OB.ENC_LTSV, "ltsv_extra"),
                            "rfc5424": (RFC5424Encoder, OB.ENC_RFC5424, None), "rfc3164": (RFC3164Encoder, OB.ENC_RFC3164, None),
                            "passthrough": (PassthroughEncoder, OB.ENC_PASSTHROUGH, None)}[enc]
    extra = {"_k1_b": "replaced", "host": "forced", "Zeta": "capital first", "_a": "x\"y\\z\n\t:1", "_counter_u64": "shadow"} if extra_key else None
    prepend = "2026-09-23T10:11Z " if enc in ("rfc3164", "passthrough") else None
    now_ts = 1438859724.638
    seen_err = set()
    for merger, om in (("none", OB.MERGE_NONE), ("line", OB.MERGE_LINE), ("nul", OB.MERGE_NUL), ("syslen", OB.MERGE_SYSLEN)):
        if merger in ("nul", "line") and enc not in ("gelf", "rfc5424"):
            continue  # the merger code is shared: all four with two encoders, none + syslen with every encoder
        e = cls({"output": {extra_key: extra}} if extra_key else None, merger=merger, prepend=prepend)
        d_out, d_off, d_st = e.encode_device(dec, d_bytes, d_offsets, len(lines), tables, now_ts=now_ts, want_status=True)
        torch.cuda.synchronize()
        out, off, st = d_out.cpu().numpy(), d_off.cpu().numpy().astype(np.uint64), d_st.cpu().numpy()
        oblob, ooffs, ost = oracle.decode_encode_batch(fmt, oenc, om, data, offsets, cfg, extra=extra, prepend=prepend, now_ts=now_ts)
        bad = np.flatnonzero(off != ooffs)
        if len(bad) or not np.array_equal(out, oblob):
            i = max(int(bad[0]) - 1, 0) if len(bad) else int(np.searchsorted(ooffs, np.flatnonzero(out != oblob)[0], side="right") - 1)
            a, b = out[int(off[i]):int(off[i + 1])].tobytes(), oblob[int(ooffs[i]):int(ooffs[i + 1])].tobytes()
            raise AssertionError(f"{merger}: line {i}: {lines[i][:160]!r}\n  gpu    {a!r}\n  oracle {b!r}")
        assert np.array_equal(np.minimum(st, 2), ost), "encode status differs"
        seen_err |= {e.error_string(int(s)) for s in np.unique(st) if s >= 2}
        if (src, enc) != ("gelf", "passthrough"):  # GELF records rarely carry full_message: "Cannot output empty raw message"
            assert (np.diff(ooffs.astype(np.int64)) > 0).sum() > 0.9 * len(lines)
    if enc == "rfc5424" and src in ("rfc5424", "gelf"):
        assert "Failed to parse date" in seen_err and "Failed to parse date as Rfc3339 format" in seen_err
    if enc == "rfc3164" and src in ("rfc5424", "gelf"):
        assert "Failed to parse unix timestamp in RFC3164 encoder" in seen_err
    if enc == "passthrough" and src == "gelf":
        assert "Cannot output empty raw message" in seen_err


# ---------------------------------------------------------------------------------------------
# RFC3164 decoder: rfc3164_decoder.rs:31-213
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def r3164(oracle):
    from flowgger_amd import tzdb

    oracle.set_rfc3164(RFC3164_YEAR, tzdb.default_table())
    return RFC3164Decoder(RFC3164_CONFIG)


def test_rfc3164_corpus_and_fuzz_bit_exact(r3164, oracle):
    """both forms, zones, the year fallback, every error string and the input on which the reference panics: host-buffer
    and device paths == the oracle; then the same lines as newline-framed stream through GPU framing + decode"""
    from test_rfc3164_cpu import fuzz_lines

    lines = [v["line"].encode() for v in RFC3164_VECTORS] + synth.rfc3164_lines(30_000) + fuzz_lines(30_000, 23)
    lines += [b"", b"<", b"<>", b"Aug", b"a: b: ", b": : ", b"Aug 6 11:15:24 " + b"h" * 9000 + b" long hostname token", "Aug 6 11:15:24 h é　x".encode()]
    both_paths(r3164, oracle, lines)
    clone = r3164.clone_boxed()            # decoder/mod.rs:29-36: the clone carries year + zone table
    both_paths(clone, oracle, lines[:500])


def test_rfc3164_long_lines_take_the_global_path(r3164, oracle):
    rng = np.random.default_rng(5)
    lines = []
    for i in range(600):
        body = " ".join(synth._WORDS[int(k)] for k in rng.integers(0, len(synth._WORDS), int(rng.integers(1, 3000))))
        lines.append((("<%d>" % (i % 192)) + "Aug %2d 11:15:24 Europe/Paris host%d " % (1 + i % 28, i) + body).encode())
    both_paths(r3164, oracle, lines)


def test_rfc3164_frames(r3164, oracle):
    import torch

    from flowgger_amd import _lib as L

    lines = synth.rfc3164_lines(20_000)
    lines[7] = lines[7] + b"\r"
    lines[9] = b"Aug  6 11:15:24 h \xff\xfe not utf-8"
    stream = b"".join(ln + b"\n" for ln in lines)
    raw = torch.frombuffer(bytearray(stream + b"\0" * 32), dtype=torch.uint8).cuda()[:len(stream)]
    d_off, d_bad, nf = r3164.frame_device(raw, L.FG_FRAME_LINE)
    assert nf == len(lines)
    from flowgger_amd.tables import DeviceTables

    tables = DeviceTables(nf, 16, raw.device)
    r3164.decode_frames_device(raw, d_off, nf, tables, L.FG_FRAME_LINE, d_bad)
    torch.cuda.synchronize()
    off = d_off[:nf + 1].cpu().numpy().astype(np.uint64)
    data = np.frombuffer(stream, np.uint8)
    blob, offs = tables.to_host().serialize(r3164.fmt, data, off)
    good = [ln[:-1] if ln.endswith(b"\r") else ln for ln in lines]
    gdata, goffs = synth.pack(good)
    oblob, ooffs = oracle.decode_batch(RFC3164, gdata, goffs)
    from flowgger_amd.record import parse_canonical

    for i in (0, 7, 8, 9, 10, len(lines) - 1):
        got = parse_canonical(blob[int(offs[i]):int(offs[i + 1])].tobytes())
        if i == 9:
            assert str(got) == "Invalid UTF-8 input"
        else:
            assert blob[int(offs[i]):int(offs[i + 1])].tobytes() == oblob[int(ooffs[i]):int(ooffs[i + 1])].tobytes(), (i, got)
    st = tables.to_host().status
    ost = np.array([oblob[int(ooffs[i])] for i in range(len(lines))])
    # every other line: same verdict (the oracle is handed line 9's bytes although they are not a &str: it has no UTF-8 check,
    # the splitter rejects the line before decode)
    keep = np.arange(len(lines)) != 9
    assert np.array_equal((st != 0)[keep], (ost != 0)[keep]) and st[9] == L.FG_ST_BAD_UTF8


@pytest.mark.parametrize("enc", ["gelf", "ltsv", "rfc5424", "rfc3164", "passthrough"])
def test_encoders_on_rfc3164_records(r3164, oracle, enc):
    """records decoded from BSD-syslog lines (msg = re-joined tokens) through every encoder + the syslen merger"""
    import torch

    import oracle_binding as OB
    from flowgger_amd import GelfEncoder, LTSVEncoder, PassthroughEncoder, RFC3164Encoder, RFC5424Encoder
    from test_rfc3164_cpu import fuzz_lines

    lines = synth.rfc3164_lines(8_000) + fuzz_lines(4_000, 31) + ["Aug 6 11:15:24 h  a\t\tb 　 c\"d\\e ".encode()]
    data, offsets = synth.pack(lines)
    tables, d_bytes, d_offsets = device_path(r3164, data, offsets)
    cls, oenc = {"gelf": (GelfEncoder, OB.ENC_GELF), "ltsv": (LTSVEncoder, OB.ENC_LTSV), "rfc5424": (RFC5424Encoder, OB.ENC_RFC5424),
                 "rfc3164": (RFC3164Encoder, OB.ENC_RFC3164), "passthrough": (PassthroughEncoder, OB.ENC_PASSTHROUGH)}[enc]
    e = cls(None, merger="syslen")
    d_out, d_off = e.encode_device(r3164, d_bytes, d_offsets, len(lines), tables)
    torch.cuda.synchronize()
    out, off = d_out.cpu().numpy(), d_off.cpu().numpy().astype(np.uint64)
    oblob, ooffs, _ = oracle.decode_encode_batch(RFC3164, oenc, OB.MERGE_SYSLEN, data, offsets)
    bad = np.flatnonzero(off != ooffs)
    if len(bad) or not np.array_equal(out, oblob):
        i = max(int(bad[0]) - 1, 0) if len(bad) else int(np.searchsorted(ooffs, np.flatnonzero(out != oblob)[0], side="right") - 1)
        a, b = out[int(off[i]):int(off[i + 1])].tobytes(), oblob[int(ooffs[i]):int(ooffs[i + 1])].tobytes()
        raise AssertionError(f"line {i}: {lines[i][:160]!r}\n  gpu    {a!r}\n  oracle {b!r}")
    assert (np.diff(ooffs.astype(np.int64)) > 0).sum() > 0.5 * len(lines)


# ---------------------------------------------------------------------------------------------
# fg_transcode_batch: the whole handle_line of a batch with host buffers (BASELINE configs[0] through the C ABI)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("src,enc,merger", [("rfc5424", "gelf", "line"), ("rfc5424", "rfc5424", "syslen"), ("ltsv", "ltsv", "line"),
                                            ("gelf", "gelf", "nul"), ("rfc3164", "rfc3164", "line"), ("rfc5424", "passthrough", "none")])
def test_transcode_batch_equals_the_reference_pipeline(rfc, r3164, oracle, src, enc, merger):
    """host buffers in -> (H2D, decode, encode, frame, D2H of the bytes only) -> exactly the oracle's
    decode -> Encoder::encode -> Merger::frame stream, with the decoder / encoder verdict per line"""
    import oracle_binding as OB
    from flowgger_amd import GelfEncoder, LTSVEncoder, PassthroughEncoder, Pipeline, RFC3164Encoder, RFC5424Encoder

    if src == "rfc3164":
        dec, fmt, cfg, lines = r3164, RFC3164, None, synth.rfc3164_lines(6_000)
    else:
        dec, fmt, cfg, lines = _encode_corpus(src, rfc)
    data, offsets = synth.pack(lines)
    cls, oenc = {"gelf": (GelfEncoder, OB.ENC_GELF), "ltsv": (LTSVEncoder, OB.ENC_LTSV), "rfc5424": (RFC5424Encoder, OB.ENC_RFC5424),
                 "rfc3164": (RFC3164Encoder, OB.ENC_RFC3164), "passthrough": (PassthroughEncoder, OB.ENC_PASSTHROUGH)}[enc]
    om = {"none": OB.MERGE_NONE, "line": OB.MERGE_LINE, "nul": OB.MERGE_NUL, "syslen": OB.MERGE_SYSLEN}[merger]
    now_ts = 1438859724.638
    pipe = Pipeline(dec, cls(None, merger=merger))
    oblob, ooffs, ost = oracle.decode_encode_batch(fmt, oenc, om, data, offsets, cfg, extra=None, prepend=None, now_ts=now_ts)
    oracle_dec, odec_offs = oracle.decode_batch(fmt, data, offsets, cfg)
    for rep in range(2):  # the second call reuses the grown buffers
        r = pipe.run_packed(data, offsets, now_ts=now_ts)
        assert r.n == len(lines) and r.consumed == int(offsets[-1])
        assert np.array_equal(r.out_offsets, ooffs) and np.array_equal(r.out, oblob)
        assert np.array_equal(np.minimum(r.enc_status, 2), ost)
        # decoder verdict per line: Ok exactly where the oracle's Record is Ok
        assert np.array_equal(r.dec_status != 0, oracle_dec[odec_offs[:-1].astype(np.int64)] != 0)
    # an empty batch and a one-line batch
    e = pipe.run_packed(np.zeros(16, np.uint8), np.zeros(1, np.uint64))
    assert e.n == 0 and len(e.out) == 0
    d1, o1 = synth.pack(lines[:1])
    one = pipe.run_packed(d1, o1, now_ts=now_ts)
    assert one.out.tobytes() == oblob[:int(ooffs[1])].tobytes()


def test_transcode_stream_chunks_concatenate(rfc, oracle):
    """raw "\\n" stream cut into arbitrary chunks -> GPU framing + UTF-8 check + decode + GELF encode + line merger:
    the concatenated output equals the oracle's pipeline over the valid frames (stdin -> rfc5424 -> gelf, configs[0])"""
    import oracle_binding as OB
    from flowgger_amd import GelfEncoder, Pipeline
    from flowgger_amd import _lib as L

    lines = synth.rfc5424_lines(20_000, cfg=2) + synth.rfc5424_lines(1_000, cfg=4, sd=True)
    tort = _utf8_torture()
    raw = b"".join((ln + (b" " + tort[i % len(tort)] if i % 11 == 4 else b"")) + (b"\r\n" if i % 4 == 0 else b"\n")
                   for i, ln in enumerate(lines)) + b"<13>1 2015-08-05T15:53:45Z h a p m - unterminated last line"
    ref = _frames_reference(raw, "line")
    good = [r[2] for r in ref if r[3]]
    gdata, goffs = synth.pack(good)
    oblob, ooffs, ost = oracle.decode_encode_batch(RFC5424, OB.ENC_GELF, OB.MERGE_LINE, gdata, goffs, None, extra=None, prepend=None, now_ts=0.0)
    pipe = Pipeline(rfc, GelfEncoder(None, merger="line"))
    rng = np.random.default_rng(5)
    out, n_frames, n_bad, pos, carry = [], 0, 0, 0, b""
    while True:
        step = int(rng.integers(1, 900_000))
        chunk = carry + raw[pos:pos + step]
        pos += step
        final = pos >= len(raw)
        r = pipe.run_stream(chunk, L.FG_FRAME_LINE, final=final)
        out.append(r.out.tobytes())
        n_frames += r.n
        n_bad += int((r.dec_status == L.FG_ST_BAD_UTF8).sum())
        if r.n:
            assert int(r.frame_offsets[-1]) == r.consumed or final
        carry = chunk[r.consumed:]
        if final:
            assert carry == b""
            break
    assert n_frames == len(ref) and n_bad == sum(1 for r in ref if not r[3])
    assert b"".join(out) == oblob.tobytes()


@pytest.mark.parametrize("fmt,framing", [("rfc5424", "pipe-line"), ("gelf", "pipe-nul"), ("rfc5424", "pipe-syslen")])
def test_cpp_transcoding_splitter_is_the_reference_pipeline(tmp_path, oracle, fmt, framing):
    """fg::TranscodingSplitter (C++): raw stream -> the GELF stream the output writes + the reference's stderr lines,
    i.e. BASELINE configs[0] (input -> decoder -> gelf encoder -> output) with every stage on the GPU; syslen framing
    ("<len> " prefixes, syslen_splitter.rs) is hopped through on the host, the messages go to the GPU packed"""
    import subprocess
    from pathlib import Path

    import oracle_binding as OB

    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "host_mirror_test"
    subprocess.run(["g++", "-std=c++17", "-O1", str(root / "tests/native/host_mirror_test.cpp"), "-o", str(exe),
                    f"-L{root / 'flowgger_amd'}", "-lfg_hip", f"-Wl,-rpath,{root / 'flowgger_amd'}",
                    "-L/opt/rocm/lib", "-lamdhip64"], check=True)
    code = {"rfc5424": RFC5424, "gelf": GELF}[fmt]
    lines = synth.rfc5424_lines(3000, cfg=4, sd=True) + synth.rfc5424_lines(3000, cfg=2) if fmt == "rfc5424" else synth.gelf_lines(3000)
    syslen = framing == "pipe-syslen"
    term = b"\n" if framing == "pipe-line" else b"\0"
    lines = [ln for ln in lines if b"\n" not in ln and b"\0" not in ln]
    bad_utf8 = b"<13>1 2015-08-05T15:53:45Z h a p m - \xff\xfe"
    if syslen:  # the reference unwrap()-panics on invalid UTF-8 here: not in the stream; it ends with an unreadable length
        raw = b"".join(str(len(ln)).encode() + b" " + ln for ln in lines) + b"12x "
    else:
        raw = b"".join(ln + term for ln in lines[:100]) + bad_utf8 + term + b"".join(ln + term for ln in lines[100:-1]) + lines[-1]
    f = tmp_path / "in.bin"
    f.write_bytes(raw)
    p = subprocess.run([str(exe), fmt, framing, str(f), "30011"], capture_output=True)
    assert p.returncode == 0, p.stderr[-2000:]
    data, offsets = synth.pack(lines)
    extra = {"_site": "dc1", "a_first": "x\"y"}
    om = OB.MERGE_SYSLEN if syslen else OB.MERGE_LINE if term == b"\n" else OB.MERGE_NUL
    oblob, ooffs, ost = oracle.decode_encode_batch(code, OB.ENC_GELF, om, data, offsets, None, extra=extra, prepend=None, now_ts=1438859724.638)
    assert p.stdout == oblob.tobytes()
    want_err = []
    for i, ln in enumerate(lines):
        if i == 100 and not syslen:
            want_err.append("Invalid UTF-8 input")
        c = oracle.decode(code, ln, None)
        if c[0] != 0:
            t = ln.decode("utf-8", "replace").strip()
            if not (framing == "pipe-nul" and t == ""):
                want_err.append(f"{c[5:].decode()}: [{t}]")
    if syslen:
        want_err.append("Can't read message's length")
    assert p.stderr.decode("utf-8", "replace").splitlines() == want_err


def test_encode_offsets_at_scale(rfc):
    """600 K lines = more 64-line blocks than one round of the offset scan holds: the output must be one gap-free stream in
    which every message ends with the merger's "\\n" and no other byte is a raw newline (GELF escapes them) -- a
    size-independent check of count -> block scan -> line offsets -> write"""
    import torch

    from flowgger_amd import GelfEncoder

    lines = synth.rfc5424_lines(600_000, cfg=2)
    data, offsets = synth.pack(lines)
    tables, d_bytes, d_offsets = device_path(rfc, data, offsets)
    d_out, d_off, d_st = GelfEncoder(None, merger="line").encode_device(rfc, d_bytes, d_offsets, len(lines), tables, want_status=True)
    torch.cuda.synchronize()
    sizes = d_off[1:] - d_off[:-1]
    assert int(d_off[0]) == 0 and int(d_off[-1]) == d_out.numel() and bool((sizes >= 0).all())
    produced = sizes > 0
    assert bool((produced == (d_st == 0)).all())
    ends = (d_off[1:][produced] - 1).long()
    assert bool((d_out[ends] == 10).all())
    assert int((d_out == 10).sum()) == int(produced.sum())
    # and the first / last messages are the oracle-checked ones of the small tests: valid JSON objects
    first = bytes(d_out[:int(d_off[1])].cpu().numpy())
    assert first.startswith(b"{") and first.endswith(b"}\n")
