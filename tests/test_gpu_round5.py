"""-m gpu, round 5: the dynamic chunk dispatch (tickets), small batches, and the wave-cooperative second look at lines whose head was
not enough.

* every format at the batch sizes a framer really hands over (1 .. 70 000 lines), under both dispatch forms and chunk sizes down
  to ONE line, several launches on one ctx (the ticket counter is never reset: the host's bookkeeping must agree with the device);
* RFC5424 lines whose structured data runs past the 1 KiB head (valid and broken behind the head), lines longer than the tile;
* chunk boundaries inside long lines are impossible by construction (chunks are line ranges) -- long-tail corpora at tiny chunk
  sizes exercise the ragged last group of every chunk instead.
The oracle is the checker (tests/oracle_binding.py); the product path is the C ABI.
"""
import numpy as np
import pytest

from flowgger_amd import GelfDecoder, LTSVDecoder, RFC3164Decoder, RFC5424Decoder, synth
from flowgger_amd import _lib as L
from gpu_util import assert_same, device_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle():
    import oracle_binding

    return oracle_binding.Oracle()


def check_device(dec, oracle, lines, config=None):
    data, offsets = synth.pack(lines)
    oblob, ooffs = oracle.decode_batch(dec.fmt, data, offsets, config)
    tables, _, _ = device_path(dec, data, offsets)
    blob, offs = tables.to_host().serialize(dec.fmt, data, offsets, cfg=dec._cfg)
    assert_same(blob, offs, oblob, ooffs, lines)


def ticket_ring_ok(dec):
    """the device's ticket counters == what the host's bookkeeping says they hold (fg_ticket_ring_check: 0 = all equal)"""
    return L.lib().fg_ticket_ring_check(dec._ctx)


def corpora():
    return {
        "rfc5424": (lambda: RFC5424Decoder(), None, synth.rfc5424_lines(70_000, cfg=2)),
        "rfc5424_sd": (lambda: RFC5424Decoder(), None, synth.rfc5424_lines(20_000, cfg=4, sd=True)),
        "rfc5424_long": (lambda: RFC5424Decoder(), None, synth.rfc5424_lines(6_000, cfg=5, sd=True, long_tail=True)),
        "ltsv": (lambda: LTSVDecoder(synth.LTSV_CONFIG), synth.LTSV_CONFIG, synth.ltsv_lines(40_000)),
        "ltsv_long": (lambda: LTSVDecoder(synth.LTSV_CONFIG), synth.LTSV_CONFIG, synth.ltsv_lines(6_000, long_tail=True)),
        "gelf": (lambda: GelfDecoder(), None, synth.gelf_lines(30_000)),
    }


@pytest.mark.parametrize("name", ["rfc5424", "rfc5424_sd", "rfc5424_long", "ltsv", "ltsv_long", "gelf"])
def test_small_batches_under_both_dispatch_forms(oracle, name):
    make, cfg, lines = corpora()[name]
    sizes = [1, 2, 63, 64, 65, 127, 1000, 5000, len(lines)]
    for opts in (dict(), dict(static_chunks=True), dict(chunk_lines=1), dict(chunk_lines=3, waves_per_cu=1),
                 dict(chunk_lines=100, waves_per_cu=2), dict(chunk_lines=64, static_chunks=True),
                 # (named chunks with room for a taper -- halves, quarters, eighths of a chunk at the end of the batch -- and without)
                 dict(chunk_lines=64, waves_per_cu=1), dict(chunk_lines=100, waves_per_cu=2, no_taper=True)):
        dec = make()
        dec.set_launch_opts(**opts)
        for n in sizes:
            if n > len(lines):
                continue
            # (a window that moves through the corpus: the invalid lines and the long ones land in different lanes and chunks)
            start = (n * 7) % max(1, len(lines) - n + 1)
            check_device(dec, oracle, lines[start:start + n], cfg)
        # the same ctx again, sizes interleaved: every launch draws from its own word of the ring, never reset
        for n in (5000, 1, 65, 1000):
            if n <= len(lines):
                check_device(dec, oracle, lines[:n], cfg)
        assert ticket_ring_ok(dec) == 0, f"{name} {opts}: ticket counters and host bookkeeping disagree"
        dec.close()


def test_ticket_ring_wraps(oracle):
    """more launches on one ctx than the ring has words: the slots are reused with their counters where earlier launches left them"""
    dec = RFC5424Decoder()
    lines = synth.rfc5424_lines(3000, cfg=2)
    data, offsets = synth.pack(lines)
    oblob, ooffs = oracle.decode_batch(dec.fmt, data, offsets, None)
    import torch

    from flowgger_amd.tables import DeviceTables

    dev = torch.device("cuda", dec.device)
    d_bytes = torch.cat([torch.from_numpy(data[: int(offsets[-1])]).to(dev), torch.zeros(32, dtype=torch.uint8, device=dev)])
    d_offsets = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    tables = DeviceTables(len(lines), 4096, dev)
    for k in range(2100):
        dec.decode_device(d_bytes, d_offsets, tables)
    torch.cuda.synchronize(dev)
    blob, offs = tables.to_host().serialize(dec.fmt, data, offsets, cfg=dec._cfg)
    assert_same(blob, offs, oblob, ooffs, lines)
    assert ticket_ring_ok(dec) == 0


def long_sd_lines(rng, n=400):
    """RFC5424 lines whose structured data ends well behind byte 1024 (the staged head), with long messages behind it: valid ones,
    and ones that are broken BEHIND the head (so that the verdict needs bytes the head does not hold)."""
    pool = synth._text_pool(rng, 1 << 16)
    out = []
    for i in range(n):
        npairs = int(rng.integers(28, 70))
        pairs = []
        for k in range(npairs):
            o = int(rng.integers(0, len(pool) - 64))
            val = pool[o:o + int(rng.integers(1, 48))].replace('"', "'").replace("\\", "/").replace("]", ")")
            if rng.random() < 0.1:
                val += '\\"x\\\\y\\]z'
            pairs.append(f'key{k}_{"abcdefgh"[k % 8]}="{val}"')
        sd = "[sd@1 " + " ".join(pairs) + "]"
        if rng.random() < 0.4:
            sd += "[e2@9 a=\"b\" c=\"d\"]"
        kind = i % 8
        if kind == 5:
            sd = sd[:-1]                      # the closing bracket is missing: "Missing ] after structured data"
        elif kind == 6:
            cut = sd.rfind(' key')
            sd = sd[:cut] + " =" + sd[cut + 1:]  # a format error far behind the head
        elif kind == 7:
            sd = sd + "x"                      # garbage behind the last ']': "Malformated RFC5424 message"
        msg_len = int(np.exp(rng.uniform(np.log(8.0), np.log(9000.0))))
        o = int(rng.integers(0, len(pool) - 9100))
        tail = (" " + pool[o:o + msg_len]) if kind != 4 else ""  # kind 4: no message behind the structured data
        out.append(f"<{int(rng.integers(0, 192))}>1 2021-03-04T05:06:07.{int(rng.integers(0, 999999)):06d}Z host{i} app {i} ID{i} {sd}{tail}".encode())
    return out


@pytest.mark.parametrize("opts", [dict(), dict(force_head=True), dict(force_head=True, tile_cap=6144), dict(force_head=True, sd_walk=True),
                                  dict(force_head=True, chunk_lines=1), dict(no_head=True)])
def test_structured_data_that_runs_past_the_staged_head(oracle, opts):
    rng = np.random.default_rng(0x5424_55)
    special = long_sd_lines(rng)
    # among ordinary long-tail lines, so that a group holds both kinds
    filler = synth.rfc5424_lines(4000, cfg=5, sd=True, long_tail=True)
    lines = []
    for i, ln in enumerate(filler):
        lines.append(ln)
        if i % 10 == 3 and special:
            lines.append(special.pop())
    lines.extend(special)
    # lines longer than any tile (the byte-wise walk through global memory stays the last resort)
    lines.append(lines[5][:200] + b"x" * 70_000)
    lines.append(long_sd_lines(rng, 1)[0] + b" " + b"y" * 50_000)
    dec = RFC5424Decoder()
    dec.set_launch_opts(**opts)
    check_device(dec, oracle, lines)
    # a batch of ONLY such lines, and one line alone
    sp = long_sd_lines(np.random.default_rng(7), 130)
    check_device(dec, oracle, sp)
    check_device(dec, oracle, sp[:1])


def test_rfc3164_is_untouched_by_the_dispatch_flag(oracle):
    """RFC3164 has its own launcher (no persistent loop): the flag must be a no-op there"""
    from flowgger_amd import tzdb

    oracle.set_rfc3164(2026, tzdb.default_table())
    dec = RFC3164Decoder({"rfc3164": {"current_year": 2026}})
    dec.set_launch_opts(static_chunks=True)
    check_device(dec, oracle, synth.rfc3164_lines(3000))


def test_device_merge_equals_the_oracle_in_arrival_order(oracle):
    """fg_merge_tables_device against the ORACLE (round 4 compared it with the host merge only): the rows of every
    sub-batch are taken back out of the merged table at their arrival positions (entry slices where the merge put them) and must
    serialise to the canonical Records the oracle decodes from that sub-batch's lines, in order."""
    import torch
    from flowgger_amd import shard
    from flowgger_amd.tables import HostTables

    n = 60_000
    tag, (la, ia), (lb, ib) = synth.mixed_cfg5(n)
    decs = (RFC5424Decoder(), LTSVDecoder(synth.LTSV_CONFIG))
    cfgs = (None, synth.LTSV_CONFIG)
    dev = torch.device("cuda", decs[0].device)
    dparts, packs = [], []
    for dec, lines in zip(decs, (la, lb)):
        data, offsets = synth.pack(lines)
        tables, _, _ = device_path(dec, data, offsets, ent_cap=int(offsets[-1]) // 16 + (1 << 20))
        dparts.append(tables)
        packs.append((data, offsets, lines))
    d_index = [torch.from_numpy(ix.astype(np.int64)).to(dev) for ix in (ia, ib)]
    out, d_src = shard.merge_tables_device(decs[0], dparts, d_index)
    torch.cuda.synchronize(dev)
    got = out.to_host()
    src = d_src.cpu().numpy()
    assert np.array_equal(src, tag)
    for k, (dec, cfg, ix) in enumerate(zip(decs, cfgs, (ia, ib))):
        data, offsets, lines = packs[k]
        rows = ix.astype(np.int64)
        assert np.array_equal(np.flatnonzero(src == k), np.sort(rows))
        arrays = {}
        for name, a in got.a.items():
            if name in ("meta", "ts", "ent_first", "ent_count"):
                arrays[name] = np.ascontiguousarray(a[rows])
            elif name in ("hostname", "appname", "procid", "msgid", "msg", "full_msg"):
                arrays[name] = np.ascontiguousarray(a.reshape(-1, 2)[rows]).reshape(-1)
            else:
                arrays[name] = a  # the merged entry columns, whole: ent_first points into them
        sub = HostTables(len(rows), got.ent_cap, arrays)
        blob, offs = sub.serialize(dec.fmt, data, offsets, cfg=dec._cfg)
        oblob, ooffs = oracle.decode_batch(dec.fmt, data, offsets, cfg)
        assert_same(blob, offs, oblob, ooffs, lines)


@pytest.mark.parametrize("framing", ["line", "nul"])
def test_one_pass_framing_at_scale_equals_the_oracles_splitter(oracle, framing):
    """k_frame_onepass on a ~300 MB stream against fgo_frame -- ALL frames and verdicts (round 4 compared with the
    classic kernels and a 1-in-37 Python sample): UTF-8 damage, empty frames, frames longer than a 128 KiB tile, CRLF, an
    unterminated tail."""
    import torch

    rng = np.random.default_rng(0x0F4B)
    dec = RFC5424Decoder()
    dev = torch.device("cuda", dec.device)
    delim = b"\n" if framing == "line" else b"\0"
    base = synth.rfc5424_lines(120_000, cfg=2)
    dmg = [b"\xff", b"\xc3", b"\xe2\x82", b"\xf0\x9f\x98", b"\xed\xa0\x80", b"\xc0\xaf", b"caf\xc3\xa9 \xe2\x82\xac ok", b"\xf4\x90\x80\x80"]
    pieces = []
    for i, ln in enumerate(base):
        if i % 53 == 7:
            ln = ln + b" " + dmg[(i // 53) % len(dmg)]
        if i % 997 == 0:
            ln = b""
        if i % 20011 == 5:
            ln = ln + b" " + b"x" * int(rng.integers(17_000, 300_000))
        if i % 101 == 3 and framing == "line":
            ln = ln + b"\r"
        pieces.append(ln + delim)
    raw = b"".join(pieces) * 9 + b"unterminated tail \xe2\x82"
    assert len(raw) > 256 << 20
    host = np.frombuffer(raw, np.uint8)
    d_bytes = torch.cat([torch.from_numpy(host.copy()), torch.zeros(32, dtype=torch.uint8)]).to(dev)
    fr = L.FG_FRAME_LINE if framing == "line" else L.FG_FRAME_NUL
    d_offsets, d_bad, n = dec.frame_device(d_bytes[:len(raw)], fr)
    got_off = d_offsets[:n + 1].cpu().numpy().astype(np.uint64)
    got_bad = d_bad[:n].cpu().numpy().astype(np.uint8)
    starts, ends, valid = oracle.frame_arrays(host, framing)
    assert n == len(starts) == len(pieces) * 9 + 1
    assert np.array_equal(got_off[:-1], starts) and np.array_equal(got_off[1:], ends) and int(got_off[-1]) == len(raw)
    want_bad = (1 - valid).astype(np.uint8)
    assert np.array_equal(got_bad, want_bad), f"first UTF-8 verdict mismatch at frame {int(np.flatnonzero(got_bad != want_bad)[0])}"


@pytest.mark.parametrize("framing", ["line", "nul"])
def test_one_pass_framing_falls_back_when_a_tile_never_publishes(oracle, framing):
    """FG_LO_FRAME_SELFTEST_STALL: the second tile of the one-pass scan withholds its descriptor, so every tile behind it gives up
    within the spin bound, the total comes back as the abort sentinel and the call must deliver the classic kernels' result -- through
    fg_frame_device and through the sliced host path (the fall-back had never run on a GPU)."""
    import ctypes as C
    import time

    import torch

    dec = RFC5424Decoder()
    dev = torch.device("cuda", dec.device)
    delim = b"\n" if framing == "line" else b"\0"
    lines = synth.rfc5424_lines(8000, cfg=2)
    raw = b"".join(ln + delim for ln in lines) + b"tail without a terminator"
    assert len(raw) > 8 * 128 * 1024
    host = np.frombuffer(raw, np.uint8)
    starts, ends, valid = oracle.frame_arrays(host, framing)
    d_bytes = torch.cat([torch.from_numpy(host.copy()), torch.zeros(32, dtype=torch.uint8)]).to(dev)
    fr = L.FG_FRAME_LINE if framing == "line" else L.FG_FRAME_NUL
    for stall in (False, True):
        dec.set_launch_opts(frame_selftest_stall=stall)
        t0 = time.perf_counter()
        d_offsets, d_bad, n = dec.frame_device(d_bytes[:len(raw)], fr)
        dt = time.perf_counter() - t0
        got = d_offsets[:n + 1].cpu().numpy().astype(np.uint64)
        assert n == len(starts) and np.array_equal(got[:-1], starts) and np.array_equal(got[1:], ends), f"stall={stall}"
        assert np.array_equal(d_bad[:n].cpu().numpy(), 1 - valid)
        assert dt < 5.0, f"the spin bound took {dt:.1f} s"
    # the sliced host path under the same stall
    lib = L.lib()
    pad = np.concatenate([host, np.zeros(64, np.uint8)])
    st, po, nf, cons = L.fg_tables(), C.c_void_p(), C.c_uint64(), C.c_uint64()
    L.check(lib.fg_frame_decode_batch(dec._ctx, dec.fmt, fr, pad.ctypes.data, host.size, 1, C.byref(st), C.byref(po), C.byref(nf), C.byref(cons)),
            "fg_frame_decode_batch")
    n = int(nf.value)
    offs = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), (n + 1,)).copy()
    assert n == len(starts) and np.array_equal(offs[:-1], starts)
    dec.set_launch_opts()


def test_bench_threads_launcher_runs_two_ranks_in_one_process():
    """bench.py --gpus 2 --launcher threads (the fall-back when torch.distributed.run cannot be used on the
    driver's node): two ranks as two threads of one process, each with its own ctx -- here both on device 0 --, barriers and
    max-over-ranks through thread barriers; rank 0 prints one JSON line for the whole job."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--launcher", "threads", "--thread-devices", "0,0",
                        "--tile-lines", "50000", "--reps", "4", "--steps", "3", "--warmup", "1", "--no-legs", "--no-mix",
                        "--no-cpu-baseline", "--no-e2e", "--no-calib"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"]["n"] == 2 and d["ranks"]["launcher"].startswith("threads")
    assert d["config"]["lines_per_gpu"] == 200_000 and d["value"] > 0
