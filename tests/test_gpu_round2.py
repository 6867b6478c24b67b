"""GPU: the boundary's concurrency contract and the bench launcher (round 2).

  * one decoder clone per connection thread (`Send`, not `Sync`): src/flowgger/input/tcp/tcp_input.rs:39-47,
    decoder/mod.rs:29-36 -- N host threads, each with its own fg_clone, decode different batches concurrently on one
    device; every result must equal the oracle's
  * `python bench.py --gpus N` launches its own ranks: the self-spawn path is exercised with one rank"""
import json
import subprocess
import sys
import threading
from pathlib import Path

import numpy as np
import pytest

from flowgger_amd import GelfDecoder, LTSVDecoder, RFC3164Decoder, RFC5424Decoder, synth, tzdb
from golden.reference_vectors import GELF, LTSV, RFC3164, RFC5424

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_eight_threads_each_with_its_own_clone(oracle):
    oracle.set_rfc3164(2026, tzdb.default_table())
    protos = {
        RFC5424: (RFC5424Decoder(), None),
        LTSV: (LTSVDecoder(synth.LTSV_CONFIG), synth.LTSV_CONFIG),
        GELF: (GelfDecoder(), None),
        RFC3164: (RFC3164Decoder({"rfc3164": {"current_year": 2026}}), None),
    }

    def corpus(fmt, seed):
        n = 20_000 + 1000 * seed
        if fmt == RFC5424:
            return synth.rfc5424_lines(n, cfg=40 + seed, sd=bool(seed % 8 >= 4))  # (cfg only seeds the generator)
        if fmt == LTSV:
            return synth.ltsv_lines(n, cfg=50 + seed)
        if fmt == GELF:
            return synth.gelf_lines(n, cfg=60 + seed)
        return synth.rfc3164_lines(n, cfg=70 + seed)

    jobs = []
    for k in range(8):
        fmt = (RFC5424, LTSV, GELF, RFC3164)[k % 4]
        proto, cfg = protos[fmt]
        data, offsets = synth.pack(corpus(fmt, k))
        want = oracle.decode_batch(fmt, data, offsets, cfg)
        jobs.append((proto.clone_boxed(), fmt, data, offsets, want))  # the clone is made on the spawning thread, as tcp_input does
    errors = []
    start = threading.Barrier(len(jobs))

    def worker(k):
        dec, fmt, data, offsets, (oblob, ooffs) = jobs[k]
        try:
            start.wait()
            for _ in range(6):  # several batches per "connection": buffers of a ctx are reused across calls
                tab = dec.decode_packed(data, offsets)
                blob, offs = tab.serialize(fmt, data, offsets, cfg=dec._cfg)
                if not (np.array_equal(offs, ooffs) and np.array_equal(blob, oblob)):
                    errors.append(f"thread {k} (format {fmt}): result differs from the oracle")
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {k}: {e!r}")

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    # the prototypes are still usable after their clones are gone
    for dec, *_ in jobs:
        dec.close()
    proto, _ = protos[RFC5424]
    assert proto.decode("<13>1 2015-08-05T15:53:45Z h a p m - x").msg == "x"


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 1 --spawn`: the torch.distributed.run self-launch that `--gpus N>1` takes when no launcher
    set WORLD_SIZE, with one rank (this box has one GPU)."""
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--spawn", "--steps", "2", "--warmup", "1", "--tile-lines", "50000",
           "--reps", "4", "--no-cpu-baseline", "--no-e2e"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0
    assert out["ranks"]["launcher"].startswith("self") and out["ranks"]["n"] == 1
    assert out["config"]["lines_per_gpu"] == 200000


@pytest.mark.parametrize("src,enc,merger", [("rfc5424", "gelf", "line"), ("sd", "rfc5424", "syslen"), ("gelf", "gelf", "nul")])
def test_transcode_large_batch_is_sliced_over_two_streams(oracle, src, enc, merger):
    """fg_transcode_batch on a large batch: upload -> decode -> encode -> download in slices (rounds 2-5: ~32 MiB slices on two lanes;
    round 6: slices that grow from 4 MiB, one stream per direction of the link).  Same bytes as the one-piece path (fg_launch_opts: FG_LO_TRANSCODE_ONE_PIECE) and as the oracle's decode -> encode -> merger."""
    import oracle_binding as OB
    from flowgger_amd import GelfEncoder, Pipeline, RFC5424Encoder

    if src == "rfc5424":
        dec, fmt, lines = RFC5424Decoder(), RFC5424, synth.rfc5424_lines(330_000, cfg=2)
    elif src == "sd":
        dec, fmt, lines = RFC5424Decoder(), RFC5424, synth.rfc5424_lines(160_000, cfg=4, sd=True)
    else:
        dec, fmt, lines = GelfDecoder(), GELF, synth.gelf_lines(260_000)
    data, offsets = synth.pack(lines)
    assert data.size > (72 << 20)
    data = np.concatenate([data, np.zeros(64, np.uint8)])
    cls, oenc = {"gelf": (GelfEncoder, OB.ENC_GELF), "rfc5424": (RFC5424Encoder, OB.ENC_RFC5424)}[enc]
    om = {"line": OB.MERGE_LINE, "nul": OB.MERGE_NUL, "syslen": OB.MERGE_SYSLEN}[merger]
    now_ts = 1438859724.638
    pipe = Pipeline(dec, cls(None, merger=merger))
    sliced = pipe.run_packed(data, offsets, now_ts=now_ts)
    again = pipe.run_packed(data, offsets, now_ts=now_ts)  # buffers already at size
    dec.set_launch_opts(transcode_one_piece=True)
    whole = pipe.run_packed(data, offsets, now_ts=now_ts)
    dec.set_launch_opts()
    for r in (sliced, again):
        assert r.n == len(lines) and r.consumed == int(offsets[-1])
        assert np.array_equal(r.out_offsets, whole.out_offsets) and np.array_equal(r.out, whole.out)
        assert np.array_equal(r.enc_status, whole.enc_status) and np.array_equal(r.dec_status, whole.dec_status)
    # the oracle on a prefix (it encodes ~250 K lines per second)
    m = 60_000
    oblob, ooffs, ost = oracle.decode_encode_batch(fmt, oenc, om, data, offsets[: m + 1], None, extra=None, prepend=None, now_ts=now_ts)
    assert np.array_equal(sliced.out_offsets[: m + 1], ooffs) and np.array_equal(sliced.out[: int(ooffs[-1])], oblob)
    assert np.array_equal(np.minimum(sliced.enc_status[:m], 2), ost)


def _frames(raw: bytes, framing: str):
    """(start, end incl. terminator, body) per frame: BufRead::lines() / split(0) (line_splitter.rs:17-25, nul_splitter.rs:18-40)."""
    delim = b"\n" if framing == "line" else b"\0"
    out, pos = [], 0
    while pos < len(raw):
        k = raw.find(delim, pos)
        end = len(raw) if k < 0 else k + 1
        body = raw[pos:end]
        if k >= 0:
            body = body[:-1]
            if framing == "line" and body.endswith(b"\r"):
                body = body[:-1]
        out.append((pos, end, body))
        pos = end
    return out


@pytest.mark.parametrize("fmt_name", ["gelf", "ltsv"])
@pytest.mark.parametrize("framing", ["line", "nul"])
def test_framed_streams_decode_like_bare_lines(oracle, fmt_name, framing):
    """GPU framing + decode of a raw stream for the decoders whose tiles carry class bitmaps: the frame terminators sit between the
    lines of a tile (GELF: they must not count as control characters of a line; a control character INSIDE a line still does),
    CRLF endings, empty frames, an unterminated tail.  Every row equals the oracle's decode of the bare line."""
    import torch
    from flowgger_amd import _lib as L
    from flowgger_amd.tables import DeviceTables

    if fmt_name == "gelf":
        dec, fmt, cfg = GelfDecoder(), GELF, None
        lines = synth.gelf_lines(6000, cfg=3)
        extra = [b'{"host":"h","a":"tab\there"}', b'{"host":"h",\t"a":1}', b'{"host":"a\rb"}', b'{"host":"h","a":"x\x01y"}', b'{"host":"h"}\r',
                 b' {"host":"h" , "b" : 1}', b'{"host":"h","s":"' + b"m" * 90 + b'"}', b'{"host":"h","s":"q\\"uote","t":"\\u00e9"}', b"{}", b"[1,2]"]
        if framing == "nul":
            extra += [b'{"host":"line1\nline2"}', b'{"host":"h",\n"a":1}', b'\n{"host":"h"}\n']
    else:
        dec, fmt, cfg = LTSVDecoder(synth.LTSV_CONFIG), LTSV, synth.LTSV_CONFIG
        lines = synth.ltsv_lines(6000, cfg=5)
        extra = [b"time:1\thost:h\tcounter:18446744073709551615\tscore:-9223372036854775808\tmean:1e308\tdone:true", b"time:1\thost:h\r",
                 b"host:h\ttime:[2000-01-01T00:00:00Z]\tmessage:m", b"time:1\thost:h\tcounter:+5", b"time:1\thost:h\tmean:.5\tscore:+7"]
    rng = np.random.default_rng(0x51a)
    delim = b"\n" if framing == "line" else b"\0"
    pieces = []
    for i, ln in enumerate(lines):
        if i % 13 == 4:
            ln = extra[(i // 13) % len(extra)]
        if framing == "line" and i % 5 == 1 and b"\n" not in ln:
            ln = ln + b"\r"
        if i % 211 == 0:
            ln = b""
        pieces.append(ln + delim)
    for tail in (b"", b'{"host":"tail"}' if fmt_name == "gelf" else b"time:1\thost:tail"):
        raw = b"".join(pieces) + tail
        ref = _frames(raw, framing)
        dev = torch.device("cuda", dec.device)
        d_bytes = torch.cat([torch.frombuffer(bytearray(raw), dtype=torch.uint8), torch.zeros(32, dtype=torch.uint8)]).to(dev)
        d_raw = d_bytes[:len(raw)]
        fr = L.FG_FRAME_LINE if framing == "line" else L.FG_FRAME_NUL
        d_offsets, d_bad, n = dec.frame_device(d_raw, fr)
        assert n == len(ref)
        offs = d_offsets[:n + 1].cpu().numpy().astype(np.uint64)
        assert np.array_equal(offs[:-1], np.array([r[0] for r in ref], np.uint64))
        tables = DeviceTables(n, len(raw) // 8 + 1024, dev)
        dec.decode_frames_device(d_raw, d_offsets, n, tables, fr, d_bad)
        torch.cuda.synchronize(dev)
        host = tables.to_host()
        bare = [r[2] for r in ref]
        gdata, goffs = synth.pack(bare)
        oblob, ooffs = oracle.decode_batch(fmt, gdata, goffs, cfg)
        blob, boffs = host.serialize(fmt, np.frombuffer(raw, np.uint8), offs, cfg=dec._cfg)
        for i in range(n):
            a = blob[int(boffs[i]):int(boffs[i + 1])].tobytes()
            b = oblob[int(ooffs[i]):int(ooffs[i + 1])].tobytes()
            assert a == b, (i, bare[i][:100])


def test_ltsv_schema_lookup_beyond_one_batch(oracle):
    """The LDS schema mirror is matched four entries per round trip: schemas of 1, 4, 5, 9 and 40 names (beyond the 32 mirrored
    entries), names longer than the 16 mirrored bytes, names that share their first 16 bytes, every type."""
    from gpu_util import host_path_blob

    types = ["u64", "i64", "f64", "bool", "string"]
    vals = {"u64": "12345", "i64": "-77", "f64": "2.5", "bool": "true", "string": "text"}
    for n_names in (1, 4, 5, 9, 40):
        names = [f"k{j:02d}" for j in range(n_names)]
        names[0] = "a_name_longer_than_sixteen_bytes_x"
        if n_names > 4:
            names[3] = "a_name_longer_than_sixteen_bytes_y"  # same first 16 bytes, same length
            names[4] = "a_name_longer_than_sixteen_bytes_yz"
        schema = {nm: types[j % 5] for j, nm in enumerate(names)}
        config = {"input": {"ltsv_schema": schema, "ltsv_suffixes": {"u64": "_u64", "bool": "_b"}}}
        dec = LTSVDecoder(config)
        lines = []
        for r in range(300):
            parts = ["time:1385053862.3072", "host:h"]
            for j in range(r % 7, n_names, max(1, n_names // 6)):
                parts.append(f"{names[j]}:{vals[schema[names[j]]]}")
            parts.append("zzz_not_in_schema:1")
            parts.append("a_name_longer_than_sixteen_bytes_q:free text")
            if r % 50 == 49:
                parts.append(f"{names[0]}:not a number")
            lines.append("\t".join(parts).encode())
        data, offsets = synth.pack(lines)
        oblob, ooffs = oracle.decode_batch(LTSV, data, offsets, config)
        (blob, offs), _ = host_path_blob(dec, data, offsets)
        assert np.array_equal(offs, ooffs)
        assert blob[:int(offs[-1])].tobytes() == oblob[:int(ooffs[-1])].tobytes()


def test_ltsv_register_parsers_match_the_byte_wise_chain(oracle):
    """The LTSV kernel decides the everyday spellings of `time` (float, RFC3339, both "English" forms) and of typed values from
    registers and leaves everything else to the byte-wise parsers: sweep the boundaries of what the register forms accept --
    every value and every error must equal the oracle's (ltsv_decoder.rs:116-199, 224-267)."""
    from gpu_util import host_path_blob

    rng = np.random.default_rng(0x7157)
    mon = ["Jan", "Feb", "Mar", "Apr", "May", "Jun", "Jul", "Aug", "Sep", "Oct", "Nov", "Dec"]
    ts = ["0", "-0", "1", "1.5", "-1.5", "1385053862.3072", "1438790025.637824", "9007199254740991", "9007199254740992", "9007199254740993",
          "90071992547409.93", "0.1234567890123456789", "123456789012345678.9", "1234567890123456789012", "00012.50", "1.", ".5", "+1.5",
          "1e5", "1E-3", "inf", "-inf", "nan", "NaN", "infinity", "1_0", "1..2", "1.2.3", "--1", "", " 1", "1 ", "0x10",
          "2000-01-01T00:00:00Z", "2000-01-01t00:00:00z", "2000-01-01T00:00:00+00:00", "2000-01-01T00:00:00-07:00", "2000-01-01T00:00:00+25:59",
          "2000-01-01T00:00:00+26:00", "2000-01-01T00:00:00.5Z", "2000-01-01T00:00:00.123456789Z", "2000-01-01T00:00:00.1234567891234Z",
          "2000-01-01T00:00:00.1234567890123456Z", "2000-01-01T00:00:00.12345678901234567Z", "2000-01-01T00:00:00.Z", "2000-01-01T00:00:00",
          "2000-01-01T00:00:60Z", "2016-12-31T23:59:60Z", "2000-02-30T00:00:00Z", "2000-13-01T00:00:00Z", "1969-12-31T23:59:59Z", "0000-01-01T00:00:00Z",
          "9999-12-31T23:59:59Z", "2000-01-01 00:00:00Z", "2000-01-01T24:00:00Z", "2000-1-01T00:00:00Z", "2000-01-01T00:00:00+0700", "2000-01-01T00:00:00ZZ",
          "5/Oct/2000:13:55:36 -0700", "05/Oct/2000:13:55:36 -0700", "15/Oct/2000:13:55:36 +0000", "15/Oct/2000:13:55:36.5 +0130",
          "15/Oct/2000:13:55:36.123456 +0130", "15/Oct/2000:13:55:36.123456789 -2559", "15/Oct/2000:13:55:36.1234567891 -0000",
          "15/Oct/2000:13:55:36. +0000", "15/Oct/2000:13:55:36 +2600", "15/Oct/2000:13:55:36 +0060", "15/oct/2000:13:55:36 +0000",
          "15/OCT/2000:13:55:36 +0000", "15/Okt/2000:13:55:36 +0000", "0/Oct/2000:13:55:36 +0000", "32/Oct/2000:13:55:36 +0000", "29/Feb/2001:00:00:00 +0000",
          "29/Feb/2000:00:00:00 +0000", "15/Oct/+2000:13:55:36 +0000", "15/Oct/-2000:13:55:36 +0000", "15/Oct/200:13:55:36 +0000", "15/Oct/20000:13:55:36 +0000",
          "15/Oct/2000:13:55:60 +0000", "15/Oct/2000:24:00:00 +0000", "15/Oct/2000:13:55:36 0000", "15/Oct/2000:13:55:36  +0000", "15/Oct/2000:13:55:36 +000",
          "15/Oct/2000:13:55:36 +00000", "15/Oct/2000:13:55:36 +0000 ", "15/Oct/1969:23:59:59 +0000", "31/Dec/1969:23:59:59 -0001", "1/Jan/1970:00:00:00 +0000",
          "1/Jan/2514:00:00:00 +0000", "1/Jan/2515:00:00:00 +0000", "115/Oct/2000:13:55:36 +0000", "1x/Oct/2000:13:55:36 +0000", "15-Oct-2000:13:55:36 +0000"]
    for _ in range(1500):
        y, mo, d = int(rng.integers(1970, 2400)), int(rng.integers(1, 13)), int(rng.integers(1, 29))
        h, mi, s2 = int(rng.integers(0, 24)), int(rng.integers(0, 60)), int(rng.integers(0, 60))
        nd = int(rng.integers(0, 12))
        frac = ("." + "".join(str(int(x)) for x in rng.integers(0, 10, nd))) if nd else ""
        sg, oh, om = "+-"[int(rng.integers(0, 2))], int(rng.integers(0, 24)), int(rng.integers(0, 60))
        k = int(rng.integers(0, 4))
        if k == 0:
            ts.append(f"{y:04d}-{mo:02d}-{d:02d}T{h:02d}:{mi:02d}:{s2:02d}{frac}" + ("Z" if om % 2 else f"{sg}{oh:02d}:{om:02d}"))
        elif k == 1:
            ts.append(f"{d}/{mon[mo - 1]}/{y:04d}:{h:02d}:{mi:02d}:{s2:02d}{frac} {sg}{oh:02d}{om:02d}")
        elif k == 2:
            ni = int(rng.integers(1, 20))
            ts.append(("-" if nd % 3 == 0 else "") + "".join(str(int(x)) for x in rng.integers(0, 10, ni)) + frac)
        else:
            ts.append(f"{int(rng.integers(0, 2**40))}.{int(rng.integers(0, 10**6)):06d}")
    nums = ["0", "-0", "+0", "5", "-5", "+5", "007", "255", "256", "18446744073709551615", "18446744073709551616", "9223372036854775807",
            "9223372036854775808", "-9223372036854775808", "-9223372036854775809", "1.5", "-1.5", "1e3", "", " ", "12a", "1234567890123456789",
            "12345678901234567890", "123456789012345678901234", "1234567890123456789012345", "true", "false", "True", "tru", "falsE", "0.1", ".1", "1."]
    lines = []
    for i, t in enumerate(ts):
        br = "[" + t + "]" if i % 2 else t
        lines.append(f"time:{br}\thost:h\tlevel:{i % 8}".encode())
    for v in nums:
        for key in ("counter", "score", "mean", "done", "level"):
            lines.append(f"time:1\thost:h\t{key}:{v}".encode())
    data, offsets = synth.pack(lines)
    dec = LTSVDecoder(synth.LTSV_CONFIG)
    oblob, ooffs = oracle.decode_batch(LTSV, data, offsets, synth.LTSV_CONFIG)
    (blob, offs), _ = host_path_blob(dec, data, offsets)
    for i in range(len(lines)):
        a = blob[int(offs[i]):int(offs[i + 1])].tobytes()
        b = oblob[int(ooffs[i]):int(ooffs[i + 1])].tobytes()
        assert a == b, (i, lines[i])


@pytest.mark.parametrize("src,enc_name", [("rfc5424", "gelf"), ("ltsv", "gelf"), ("gelf", "ltsv"), ("rfc5424sd", "rfc5424")])
def test_encode_device_async_needs_no_host_sync(oracle, src, enc_name):
    """fg_encode_device_async queues count + scan + write and returns: same bytes, offsets and statuses as fg_encode_device; with a
    buffer that is too small the output is left untouched and the need stands in d_out_offsets[n]."""
    import torch
    from flowgger_amd import GelfEncoder, LTSVEncoder, RFC5424Encoder
    from gpu_util import device_path

    if src == "rfc5424":
        dec, lines = RFC5424Decoder(), synth.rfc5424_lines(30_000, cfg=2)
    elif src == "rfc5424sd":
        dec, lines = RFC5424Decoder(), synth.rfc5424_lines(20_000, cfg=4, sd=True)
    elif src == "ltsv":
        dec, lines = LTSVDecoder(synth.LTSV_CONFIG), synth.ltsv_lines(30_000, cfg=5)
    else:
        dec, lines = GelfDecoder(), synth.gelf_lines(30_000, cfg=3)
    enc = {"gelf": GelfEncoder, "ltsv": LTSVEncoder, "rfc5424": RFC5424Encoder}[enc_name]({"output": {"framing": "line"}})
    data, offsets = synth.pack(lines)
    tables, d_bytes, d_offsets = device_path(dec, data, offsets)
    n = len(lines)
    ref_out, ref_off, ref_st = enc.encode_device(dec, d_bytes, d_offsets, n, tables, now_ts=1.5, want_status=True)
    torch.cuda.synchronize()
    total = int(ref_off[n].item())
    assert total == ref_out.numel() and total > 0
    for hint in (0xFFFFFFFFFFFFFFFF, 0 if src == "rfc5424" else 64 * n):
        out = torch.full((total + 64,), 0xAA, dtype=torch.uint8, device=d_bytes.device)
        d_off, d_st = enc.encode_device_async(dec, d_bytes, d_offsets, n, tables, out, now_ts=1.5, ent_hint=hint)
        torch.cuda.synchronize()
        assert torch.equal(d_off, ref_off) and torch.equal(d_st, ref_st)
        assert torch.equal(out[:total], ref_out) and bool((out[total:] == 0xAA).all())
    for cap in (total - 1, total // 2, 16):
        buf = torch.full((total + 64,), 0x55, dtype=torch.uint8, device=d_bytes.device)
        d_off, d_st = enc.encode_device_async(dec, d_bytes, d_offsets, n, tables, buf[:cap], now_ts=1.5)
        torch.cuda.synchronize()
        assert int(d_off[n].item()) == total and torch.equal(d_off, ref_off)
        assert bool((buf == 0x55).all()), f"a buffer of {cap} bytes (need {total}) was written to"
