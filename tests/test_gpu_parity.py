"""GPU: parity of the HIP decode path (through the C ABI) against the CPU oracle.

Bit-exact is the bar: the canonical Record serialisation of every line (Ok and Err alike, f64
timestamps as raw IEEE bits) must be byte-identical to the oracle's."""
import numpy as np
import pytest

from flowgger_amd import GelfDecoder, LTSVDecoder, RFC5424Decoder, pack_lines, synth
from flowgger_amd.record import DecodeError
from golden.reference_vectors import DERIVED_RFC5424, GELF, LTSV, RFC5424, VECTORS
from gpu_util import assert_same, device_path, host_path_blob
from test_oracle_golden import check_vector

pytestmark = pytest.mark.gpu

DECODERS = {RFC5424: RFC5424Decoder, LTSV: LTSVDecoder, GELF: GelfDecoder}
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


@pytest.mark.parametrize("v", VECTORS, ids=[v["src"].split()[-1] for v in VECTORS])
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


def test_entry_table_overflow_is_reported(rfc):
    import torch

    lines = synth.rfc5424_lines(2000, cfg=4, sd=True)
    data, offsets = synth.pack(lines)
    tables, _, _ = device_path(rfc, data, offsets, ent_cap=100)
    used = int(tables.column("ent_used").view(torch.int64)[0].item())
    assert used > 100
    meta = tables.column("meta").view(torch.int32).cpu().numpy()
    assert ((meta & 0xFF) == 0xFE).sum() > 0


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
