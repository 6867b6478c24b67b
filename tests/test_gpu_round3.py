"""GPU: round-3 drop-in behaviour -- the framers over a live pipe (time / short-read flush, idle close), the micro-batching adapter of
the per-record callers, LTSV's stdout side effect from the kernels' row flags."""
import os
import subprocess
import time
from pathlib import Path

import numpy as np
import pytest

from flowgger_amd import LTSVDecoder, synth
from flowgger_amd.tables import tables_stdout
from gpu_util import device_path
from test_abi_cpu import LTSV_NOVALUE_LINES

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
RFC5424, LTSV = 0, 1


@pytest.fixture(scope="module")
def host_exe(tmp_path_factory):
    exe = tmp_path_factory.mktemp("host3") / "host_mirror_test"
    subprocess.run(["g++", "-std=c++17", "-O1", str(ROOT / "tests/native/host_mirror_test.cpp"), "-o", str(exe),
                    f"-L{ROOT / 'flowgger_amd'}", "-lfg_hip", f"-Wl,-rpath,{ROOT / 'flowgger_amd'}", "-L/opt/rocm/lib", "-lamdhip64"],
                   check=True)
    return exe


def read_lines(stream, n, timeout):
    """n lines from a pipe, at most `timeout` seconds: -> (lines, seconds until the last one)"""
    import select

    got, buf, t0 = [], b"", time.monotonic()
    while len(got) < n and time.monotonic() - t0 < timeout:
        r, _, _ = select.select([stream], [], [], 0.05)
        if not r:
            continue
        chunk = os.read(stream.fileno(), 1 << 16)
        if not chunk:
            break
        buf += chunk
        *full, buf = buf.split(b"\n")
        got += full
    return got, time.monotonic() - t0


@pytest.mark.parametrize("mode", ["fd-line", "fd-gpu-line", "fd-pipe-line"])
def test_ten_lines_then_a_stall_come_out_within_the_latency_bound(host_exe, oracle, mode):
    """a pipe delivers 10 lines and then stalls -- all 10 come out within the latency bound (the framers used to
    wait for 8 MiB or EOF); after `idle_timeout` of silence the reference's message is printed and the connection ends
    (line_splitter.rs:26-33).  Host framing, GPU framing and the whole-pipeline splitter."""
    lines = synth.rfc5424_lines(40, cfg=2, invalid_frac=0)
    p = subprocess.Popen([str(host_exe), "rfc5424", mode, "-", "1500", "5"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE)
    try:
        # warm-up: the first GPU call of a process loads the code object
        p.stdin.write(b"\n".join(lines[:3]) + b"\n")
        p.stdin.flush()
        warm, _ = read_lines(p.stdout, 3, 60.0)
        assert len(warm) == 3, p.stderr.read()
        p.stdin.write(b"\n".join(lines[3:13]) + b"\n" + lines[13][:20])  # 10 complete lines + the beginning of an eleventh
        p.stdin.flush()
        got, dt = read_lines(p.stdout, 10, 10.0)
        assert len(got) == 10, (len(got), dt)
        assert dt < 0.25, f"10 lines took {dt * 1e3:.0f} ms to come out of a stalled connection"
        if mode != "fd-pipe-line":
            for ln, hx in zip(lines[3:13], got):
                assert bytes.fromhex(hx.decode()) == oracle.decode(RFC5424, ln)
        # silence: the idle timeout closes the connection, the partial line is dropped like `lines()` drops it
        t0 = time.monotonic()
        rest = p.stdout.read()
        p.wait(timeout=30)
        assert 1.0 < time.monotonic() - t0 < 6.0
        assert rest == b"" and p.returncode == 0
        assert b"Client hasn't sent any data for a while - Closing idle connection" in p.stderr.read()
    finally:
        p.kill()


def test_micro_batcher_delivers_in_order_with_few_gpu_calls(host_exe, oracle, tmp_path):
    """The per-record callers' adapter (udp_input.rs:139 & co.): one push() per record, results in arrival order == the oracle's,
    errors reported at their place, and the records travel in batches (max_lines 512), not one GPU call each."""
    lines = [ln for ln in synth.rfc5424_lines(5000, cfg=4, sd=True) if b"\n" not in ln]
    lines[100] = b"<13>1 2015-08-05T15:53:45Z h a p m - \xff\xfe"
    f = tmp_path / "records.txt"
    f.write_bytes(b"\n".join(lines) + b"\n")
    t0 = time.monotonic()
    r = subprocess.run([str(host_exe), "rfc5424", "micro", str(f), "512", "50"], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = r.stdout.split(b"\n")[:-1]
    errs = r.stderr.decode().split("\n")
    want_ok, want_err = [], []
    for ln in lines:
        if ln is lines[100]:
            want_err.append("Invalid UTF-8 input")
            continue
        c = oracle.decode(RFC5424, ln)
        if c[0] == 0:
            want_ok.append(c)
        else:
            want_err.append(c[5:].decode())
    assert [bytes.fromhex(h.decode()) for h in out] == want_ok
    assert errs[: len(want_err)] == want_err
    assert "at most 512 parked" in r.stderr.decode() or "at most 511 parked" in r.stderr.decode()
    assert time.monotonic() - t0 < 60


def test_ltsv_stdout_side_effect_matches_the_reference_text(oracle):
    """ltsv_decoder.rs:99 println!("Missing value for name '{}'"): rows flagged FG_F_LTSV_NOVALUE by the kernels (both the tile
    walker and the byte-wise one), the count of a failed row, and fg_tables_stdout == what the oracle's decode prints, line by line."""
    dec = LTSVDecoder(synth.LTSV_CONFIG)
    rng = np.random.default_rng(99)
    base = synth.ltsv_lines(3000, invalid_frac=0.02)
    lines = list(LTSV_NOVALUE_LINES)
    for ln in base:
        parts = ln.split(b"\t")
        k = int(rng.integers(0, 4))
        for _ in range(k):  # drop the ':' of some parts, add empty parts
            j = int(rng.integers(0, len(parts)))
            parts[j] = parts[j].replace(b":", b"", 1) if rng.integers(0, 2) else b""
        lines.append(b"\t".join(parts))
    lines.append(b"q\t" + b"x" * 70000 + b"\ttime:1\thost:h")  # longer than any tile: the byte-wise walker
    data, offsets = synth.pack(lines)
    want = b"".join(oracle.decode_stdout(LTSV, ln, synth.LTSV_CONFIG) for ln in lines)
    assert want.count(b"Missing value") > 1000
    tables, _, _ = device_path(dec, data, offsets)
    host = tables.to_host()
    assert tables_stdout(host, LTSV, np.concatenate([data, np.zeros(16, np.uint8)]), offsets) == want
    # the Records themselves are unchanged by the flag
    blob, offs = host.serialize(LTSV, data, offsets, cfg=dec._cfg)
    oblob, ooffs = oracle.decode_batch(LTSV, data, offsets, synth.LTSV_CONFIG)
    assert np.array_equal(offs, ooffs) and np.array_equal(blob, oblob)
    # the host-buffer path and the meta-only form (what fg_transcode_batch hands back)
    tab = dec.decode_packed(data, offsets)
    assert tables_stdout(tab, LTSV, np.concatenate([data, np.zeros(16, np.uint8)]), offsets) == want
    import ctypes as C

    from flowgger_amd import _lib as L

    only_meta = L.fg_tables()
    only_meta.n = tab.n
    only_meta.meta = tab.a["meta"].ctypes.data
    pad = np.concatenate([data, np.zeros(16, np.uint8)])
    n = L.lib().fg_tables_stdout(LTSV, 0, pad.ctypes.data, offsets.ctypes.data, C.byref(only_meta), 0, tab.n, None, 0)
    buf = np.zeros(int(n) + 1, np.uint8)
    L.lib().fg_tables_stdout(LTSV, 0, pad.ctypes.data, offsets.ctypes.data, C.byref(only_meta), 0, tab.n, buf.ctypes.data, int(n))
    assert buf[: int(n)].tobytes() == want


@pytest.mark.parametrize("fmt_name,final", [("rfc5424", True), ("rfc5424", False), ("ltsv", True), ("gelf", True)])
def test_raw_stream_pipelined_over_three_streams_equals_one_piece_and_oracle(oracle, fmt_name, final):
    """fg_frame_decode_batch above 48 MiB: the chunk crosses the link in slices, each slice is framed as soon as it is there (the
    delimiter ranks continue where the slice before stopped), the frames that end in it are decoded and their rows go back while the
    next slices are still on the link.  Same tables, frame offsets and `consumed` as the one-piece path -- with "\\r\\n", invalid
    UTF-8 (also across a slice boundary), an unterminated tail -- and the oracle's Records on a prefix."""
    from flowgger_amd import GelfDecoder, RFC5424Decoder
    from flowgger_amd import _lib as L

    if fmt_name == "rfc5424":
        dec, fmt, cfg = RFC5424Decoder(), 0, None
        lines = synth.rfc5424_lines(150_000, cfg=2) + synth.rfc5424_lines(60_000, cfg=4, sd=True)
    elif fmt_name == "ltsv":
        dec, fmt, cfg, lines = LTSVDecoder(synth.LTSV_CONFIG), 1, synth.LTSV_CONFIG, synth.ltsv_lines(260_000)
    else:
        dec, fmt, cfg = GelfDecoder(), 2, None
        lines = [ln for ln in synth.gelf_lines(230_000) if b"\n" not in ln]
    rng = np.random.default_rng(7)
    parts = []
    for i, ln in enumerate(lines):
        if i % 997 == 5:
            ln = ln + b" \xe4\xb8"          # a sequence cut off by the terminator
        if i % 1499 == 7:
            ln = b"\xff" + ln
        parts.append(ln + (b"\r\n" if i % 7 == 0 else b"\n"))
    raw = b"".join(parts) + b"<13>1 2015-08-05T15:53:45Z tail without a terminator"
    assert len(raw) > (56 << 20)
    # a multi-byte sequence right across the first slice boundary (8 MiB): overwrite message text, keep the framing
    buf = bytearray(raw)
    cut = 8 << 20
    for p in (cut - 1, cut, cut + 1):
        assert buf[p] not in (0x0A, 0x0D)
    buf[cut - 1:cut + 2] = "中".encode()
    raw = bytes(buf)
    piped = dec.frame_decode_batch(raw, L.FG_FRAME_LINE, final=final)
    dec.set_launch_opts(transcode_one_piece=True)
    whole = dec.frame_decode_batch(raw, L.FG_FRAME_LINE, final=final)
    dec.set_launch_opts()
    (pt, poff, pcons), (wt, woff, wcons) = piped, whole
    assert pcons == wcons and np.array_equal(poff, woff) and pt.n == wt.n == len(lines) + (1 if final else 0)
    assert pcons == (len(raw) if final else len(raw) - len(b"<13>1 2015-08-05T15:53:45Z tail without a terminator"))
    data = np.frombuffer(raw + b"\0" * 16, np.uint8)
    pb, po = pt.serialize(fmt, data, poff, cfg=dec._cfg)
    wb, wo = wt.serialize(fmt, data, woff, cfg=dec._cfg)
    assert np.array_equal(po, wo) and np.array_equal(pb, wb)
    assert int((pt.status == 0xFD).sum()) == len([i for i in range(len(lines)) if i % 997 == 5 or i % 1499 == 7])
    # the oracle on a prefix of the frames (terminators stripped as BufRead::lines() does)
    m = 40_000
    stripped = [p[:-2] if p.endswith(b"\r\n") else p[:-1] for p in parts[:m]]
    for i in rng.integers(0, m, 3000):
        got = pb[int(po[i]):int(po[i + 1])].tobytes()
        if pt.status[i] == 0xFD:
            continue
        assert got == oracle.decode(fmt, stripped[i], cfg), i


def test_sliced_host_paths_bring_entries_back_per_slice_and_survive_an_entry_table_that_is_too_small(oracle):
    """fg_decode_batch / fg_frame_decode_batch on batches cut into slices: the entry columns come back per slice (the range between
    two values of the shared counter).  Lines made of tiny pairs need more than the first capacity (one entry per 16 input bytes):
    the sliced path must notice in the middle of the batch, take the counter's final value and decode again -- same Records as the
    oracle either way; the raw-stream path falls back to its one-piece form."""
    from flowgger_amd import RFC5424Decoder
    from flowgger_amd import _lib as L
    from gpu_util import host_path_blob

    rng = np.random.default_rng(11)
    dense = [b"<13>1 2015-08-05T15:53:45Z h a p m [x@1 " + b" ".join(b'%c="%d"' % (97 + j % 26, (i + j) % 10) for j in range(int(rng.integers(30, 60))))
             + b"] m%d" % i for i in range(90_000)]
    sparse = synth.rfc5424_lines(100_000, cfg=2)
    for lines in (sparse[:50_000] + dense * 2 + sparse[50_000:], dense * 2):
        data, offsets = synth.pack(lines)
        assert data.size > (36 << 20)
        dec = RFC5424Decoder()
        oblob, ooffs = oracle.decode_batch(RFC5424, data, offsets)
        (blob, offs), tab = host_path_blob(dec, data, offsets)
        assert np.array_equal(offs, ooffs) and np.array_equal(blob, oblob)
        total = int(tab.a["ent_count"].sum())
        assert total <= tab.ent_used and total > data.size // 16  # (the first capacity was too small)
        raw = b"\n".join(lines) + b"\n"
        assert len(raw) > (48 << 20)
        ft, foff, cons = dec.frame_decode_batch(raw, L.FG_FRAME_LINE, final=True)
        assert cons == len(raw) and ft.n == len(lines)
        fb, fo = ft.serialize(RFC5424, np.frombuffer(raw + b"\0" * 16, np.uint8), foff, cfg=dec._cfg)
        assert np.array_equal(fo, ooffs) and np.array_equal(fb, oblob)
