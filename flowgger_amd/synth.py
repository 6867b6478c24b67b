"""Deterministic synthetic corpora for the BASELINE.json configurations.

cfg1/cfg2  RFC5424, no structured data, line length ~ U[192,320] (avg 256 B)
cfg4       RFC5424 with 1-2 SD elements, ~12 pairs, avg ~512 B, 5 % escaped values
cfg3       GELF with 8 flat extra fields, shuffled key order, 5 % escaped strings
cfg5       mixed lengths, log-uniform 64 B .. 8 KiB (RFC5424 and LTSV generators take `long_tail`)
Every corpus carries a fixed tail of INVALID lines (one per reachable error string) so that
Err parity is exercised as well.  Seeds: 0x54240000 + cfg (numpy PCG64).
"""
from __future__ import annotations

import json
from typing import List

import numpy as np

SEED_BASE = 0x54240000

_WORDS = ("error warn info debug request response user session token cache miss hit db query slow fast "
          "timeout retry connect closed open read write flush sync async queue worker thread pool alloc "
          "free page fault disk net tcp udp http https tls auth login logout admin root service unit "
          "started stopped failed ok status code latency ms bytes sent received from to id key value "
          "path file dir mount volume node pod container image tag build deploy rollback canary "
          "alpha beta gamma delta 127.0.0.1 10.0.0.7 fe80::1 /var/log/app.log GET POST PUT 200 404 500").split()

_MONTHS = ["Jan", "Feb", "Mar", "Apr", "May", "Jun", "Jul", "Aug", "Sep", "Oct", "Nov", "Dec"]


def _text_pool(rng: np.random.Generator, size: int = 1 << 20) -> str:
    idx = rng.integers(0, len(_WORDS), size // 4)
    return " ".join(_WORDS[i] for i in idx)


def _len_targets(rng, n, lo, hi, long_tail):
    if long_tail:  # log-uniform 64 B .. 8 KiB
        return np.exp(rng.uniform(np.log(64.0), np.log(8192.0), n)).astype(np.int64)
    return rng.integers(lo, hi + 1, n)


def _ts_fields(rng, n):
    return dict(
        y=rng.integers(2000, 2038, n), mo=rng.integers(1, 13, n), d=rng.integers(1, 29, n),
        h=rng.integers(0, 24, n), mi=rng.integers(0, 60, n), s=rng.integers(0, 60, n),
        us=rng.integers(0, 1000000, n), tzk=rng.integers(0, 10, n), tzh=rng.integers(0, 15, n),
        tzm=rng.choice([0, 15, 30, 45], n), tzs=rng.integers(0, 2, n))


def _rfc3339(f, i) -> str:
    if f["tzk"][i] < 8:
        tz = "Z"
    else:
        tz = f"{'+-'[f['tzs'][i]]}{f['tzh'][i]:02d}:{f['tzm'][i]:02d}"
    return (f"{f['y'][i]:04d}-{f['mo'][i]:02d}-{f['d'][i]:02d}T{f['h'][i]:02d}:{f['mi'][i]:02d}:"
            f"{f['s'][i]:02d}.{f['us'][i]:06d}{tz}")


_ESC_VALUES = ['a\\"b', "back\\\\slash", "br\\]acket", 'mix \\" \\\\ \\] end', "odd\\q"]


def _sd_block(rng, pool: str) -> str:
    out = []
    for e in range(int(rng.integers(1, 3))):
        npairs = int(min(max(rng.poisson(12 if e == 0 else 4), 0), 30))
        parts = [f"sd{e}@{int(rng.integers(1, 99999))}"]
        for k in range(npairs):
            if rng.random() < 0.05:
                val = _ESC_VALUES[int(rng.integers(0, len(_ESC_VALUES)))]
            else:
                o = int(rng.integers(0, len(pool) - 64))
                val = pool[o:o + int(rng.integers(1, 40))].replace('"', "'").replace("\\", "/").replace("]", ")")
            parts.append(f'k{k}_{"abcdefgh"[k % 8]}="{val}"')
        out.append("[" + " ".join(parts) + "]")
    return "".join(out)


def rfc5424_invalid_lines() -> List[str]:
    """One line per reachable RFC5424 error string (codes 1..17)."""
    hdr = "<13>1 2015-08-05T15:53:45.637824Z host app 1234 ID7 "
    return [
        "no angle bracket at all",                        # 1 Unsupported BOM
        "\ufeffBOM but no bracket",                       # 2 The priority should be inside brackets
        "<1x3>1 2015-08-05T15:53:45Z h a p m - x",        # 3 Invalid priority
        "<13",                                            # 4 Missing version
        "<13>2 2015-08-05T15:53:45Z h a p m - x",         # 5 Unsupported version
        "<13>1",                                          # 6 Missing timestamp
        "<13>1 2015-02-30T15:53:45Z h a p m - x",         # 7 Unable to parse the date ...
        "<13>1 2015-08-05T15:53:45Z",                     # 8 Missing hostname
        "<13>1 2015-08-05T15:53:45Z h",                   # 9 Missing application name
        "<13>1 2015-08-05T15:53:45Z h a",                 # 10 Missing process id
        "<13>1 2015-08-05T15:53:45Z h a p",               # 11 Missing message id
        "<13>1 2015-08-05T15:53:45Z h a p m",             # 12 Missing message data
        hdr,                                              # 13 Missing log message
        hdr + "garbage instead of sd",                    # 14 Malformated RFC5424 message
        hdr + "[nospace]",                                # 15 Missing structured data
        hdr + '[id k= "v"] msg',                          # 16 Format error in the structured data
        hdr + '[id k="v" msg',                            # 17 Missing ] after structured data
    ]


def rfc5424_lines(n: int, cfg: int = 2, sd: bool = False, invalid_frac: float = 0.01, long_tail: bool = False,
                  lo: int = 192, hi: int = 320) -> List[bytes]:
    rng = np.random.default_rng(SEED_BASE + cfg)
    pool = _text_pool(rng)
    hosts = [f"h{i:04d}.dc{i % 7}.example.com" for i in range(1024)]
    apps = [f"app-{_WORDS[i % len(_WORDS)]}{i}" for i in range(64)]
    f = _ts_fields(rng, n)
    pri = rng.integers(0, 192, n)
    hi_ = rng.integers(0, 1024, n)
    ai = rng.integers(0, 64, n)
    pid = rng.integers(1, 65536, n)
    mid = rng.integers(0, 4000, n)
    tgt = _len_targets(rng, n, lo if not sd else 384, hi if not sd else 640, long_tail)
    po = rng.integers(0, len(pool) - 9000, n)
    invalid = rfc5424_invalid_lines()
    inv_every = int(1 / invalid_frac) if invalid_frac > 0 else 0
    out: List[bytes] = []
    for i in range(n):
        if inv_every and i % inv_every == inv_every - 1:
            out.append(invalid[(i // inv_every) % len(invalid)].encode())
            continue
        msgid = f"ID{mid[i]}" if mid[i] % 4 else "-"
        head = f"<{pri[i]}>1 {_rfc3339(f, i)} {hosts[hi_[i]]} {apps[ai[i]]} {pid[i]} {msgid} "
        if sd:
            head += _sd_block(rng, pool) + " "
        else:
            head += "- "
        need = max(int(tgt[i]) - len(head), 1)
        out.append((head + pool[po[i]:po[i] + need]).encode())
    return out


def gelf_lines(n: int, cfg: int = 3, invalid_frac: float = 0.01) -> List[bytes]:
    rng = np.random.default_rng(SEED_BASE + cfg)
    pool = _text_pool(rng)
    invalid = ['{"host": 1}', "[1,2]", '{"host":"h","timestamp":"x"}', "{not json}", '{"version":"42","host":"h"}',
               '{"host":"h","level":8}', '{"host":"h","level":"x"}', '{"host":"h","k":[1]}', '{"timestamp":1}',
               '{"host":"h","short_message":5}', '{"host":"h","full_message":5}', '{"host":"h","version":1}']
    inv_every = int(1 / invalid_frac) if invalid_frac > 0 else 0
    esc = ['quote \\" inside', "back\\\\slash", "line\\nbreak", "caf\\u00e9 é"]
    out = []
    for i in range(n):
        if inv_every and i % inv_every == inv_every - 1:
            out.append(invalid[(i // inv_every) % len(invalid)].encode())
            continue
        o = int(rng.integers(0, len(pool) - 400))

        def text(a, b):
            if rng.random() < 0.05:
                return esc[int(rng.integers(0, len(esc)))]
            oo = int(rng.integers(0, len(pool) - 200))
            return pool[oo:oo + int(rng.integers(a, b))]

        ts = f"{int(rng.integers(946684800, 2145916800))}.{int(rng.integers(0, 1000000)):06d}"
        fl = f"{int(rng.integers(0, 100000))}.{int(rng.integers(0, 10000)):04d}"
        fields = [
            '"version":"1.1"', f'"host":"h{int(rng.integers(0, 1024)):04d}.example.com"',
            f'"short_message":"{text(20, 80)}"', f'"timestamp":{ts}', f'"level":{int(rng.integers(0, 8))}',
            f'"_app":"{text(4, 12)}"', f'"_env":"{text(3, 9)}"', f'"_path":"{text(8, 30)}"',
            f'"_user_id":{int(rng.integers(0, 1 << 31))}', f'"_bytes":{int(rng.integers(0, 1 << 40))}',
            f'"_delta":-{int(rng.integers(1, 100000))}', f'"_ratio":{fl}',
            '"_flag":' + ("true", "false", "null")[int(rng.integers(0, 3))],
        ]
        rng.shuffle(fields)
        sep = ", " if o % 3 == 0 else ","
        out.append(("{" + sep.join(fields) + "}").encode())
    return out


LTSV_CONFIG = {"input": {"ltsv_schema": {"counter": "u64", "score": "i64", "mean": "f64", "done": "bool"},
                         "ltsv_suffixes": {"u64": "_u64", "f64": "_f64"}}}


def ltsv_lines(n: int, cfg: int = 5, invalid_frac: float = 0.01, long_tail: bool = False) -> List[bytes]:
    rng = np.random.default_rng(SEED_BASE + cfg + 100)
    pool = _text_pool(rng).replace("\t", " ")
    f = _ts_fields(rng, n)
    tgt = _len_targets(rng, n, 192, 320, long_tail)
    invalid = ["host:h", "time:1", "time:notatime\thost:h", "time:1\thost:h\tlevel:9", "time:1\thost:h\tlevel:x",
               "time:1\thost:h\tdone:yes", "time:1\thost:h\tmean:x", "time:1\thost:h\tscore:1.5", "time:1\thost:h\tcounter:-1"]
    inv_every = int(1 / invalid_frac) if invalid_frac > 0 else 0
    out = []
    for i in range(n):
        if inv_every and i % inv_every == inv_every - 1:
            out.append(invalid[(i // inv_every) % len(invalid)].encode())
            continue
        k = i % 4
        if k == 0:
            t = f"{int(rng.integers(946684800, 2145916800))}.{int(rng.integers(0, 1000000)):06d}"
        elif k == 1:
            t = "[" + _rfc3339(f, i) + "]"
        elif k == 2:
            t = (f"[{f['d'][i]}/{_MONTHS[f['mo'][i] - 1]}/{f['y'][i]:04d}:{f['h'][i]:02d}:{f['mi'][i]:02d}:"
                 f"{f['s'][i]:02d} {'+-'[f['tzs'][i]]}{f['tzh'][i]:02d}{f['tzm'][i]:02d}]")
        else:
            t = (f"[{f['d'][i]}/{_MONTHS[f['mo'][i] - 1]}/{f['y'][i]:04d}:{f['h'][i]:02d}:{f['mi'][i]:02d}:"
                 f"{f['s'][i]:02d}.{f['us'][i]:06d} {'+-'[f['tzs'][i]]}{f['tzh'][i]:02d}{f['tzm'][i]:02d}]")
        parts = [f"time:{t}", f"host:h{int(rng.integers(0, 1024)):04d}.example.com", f"level:{int(rng.integers(0, 8))}",
                 f"counter:{int(rng.integers(0, 1 << 40))}", f"score:{int(rng.integers(-100000, 100000))}",
                 f"mean:{int(rng.integers(0, 1000))}.{int(rng.integers(0, 100000)):05d}",
                 "done:" + ("true", "false")[int(rng.integers(0, 2))], f"req_id:{int(rng.integers(0, 1 << 30)):x}",
                 "path:/api/v1/" + _WORDS[int(rng.integers(0, len(_WORDS)))]]
        head = "\t".join(parts) + "\tmessage:"
        need = max(int(tgt[i]) - len(head), 1)
        o = int(rng.integers(0, len(pool) - 9000))
        out.append((head + pool[o:o + need]).encode())
    return out


def mixed_cfg5(n: int, invalid_frac: float = 0.01):
    """BASELINE configs[4]: a 50/50 stream of RFC5424 and LTSV lines, log-uniform 64 B .. 8 KiB, as two TAGGED sub-batches --
    the reference has one decoder per input (flowgger/mod.rs:413-422), so a mixed stream is two inputs whose lines carry
    their arrival position.  Returns (tag uint8[n], (rfc5424 lines, their positions), (ltsv lines, their positions)); the
    host-side ordered gather (fg_merge_tables) puts the decoded rows back at those positions."""
    rng = np.random.default_rng(SEED_BASE + 5 + 200)
    tag = rng.integers(0, 2, n).astype(np.uint8)
    ia, ib = np.flatnonzero(tag == 0), np.flatnonzero(tag == 1)
    la = rfc5424_lines(len(ia), cfg=5, sd=True, invalid_frac=invalid_frac, long_tail=True)
    lb = ltsv_lines(len(ib), invalid_frac=invalid_frac, long_tail=True)
    return tag, (la, ia.astype(np.uint64)), (lb, ib.astype(np.uint64))


def pack(lines: List[bytes]):
    offsets = np.zeros(len(lines) + 1, np.uint64)
    offsets[1:] = np.cumsum(np.fromiter((len(b) for b in lines), np.int64, len(lines)))
    return np.frombuffer(b"".join(lines), np.uint8).copy(), offsets


def dumps_gelf(d: dict) -> bytes:  # helper for hand-written test cases
    return json.dumps(d, separators=(",", ":")).encode()


def rfc3164_invalid_lines() -> List[str]:
    """One line per RFC3164 status (fg_error_string(FG_RFC3164, 1..7))."""
    return [
        "<13 no closing bracket",                              # 1 Malformed RFC3164 event: Invalid priority
        "<1x3>Aug  6 11:15:24 host app: message",              # 2 Invalid priority
        "neither form matches this line",                      # 3 Malformed RFC3164 event: Invalid timestamp or hostname
        "host: Aug 6: message after a two-token date",         # 4 Invalid time format
        "host: Aug 36 11:15:24: day out of range, 3 tokens",   # 5 Unable to parse RFC3164 date with year
        "host: 2020 Feb 30 11:15:24: no such day",             # 6 Unable to parse the date in RFC3164 decoder
        "<13>Aug  6 11:15:24 UTC",                             # 7 the reference panics (index out of bounds)
    ]


def rfc3164_lines(n: int, cfg: int = 6, invalid_frac: float = 0.01, lo: int = 128, hi: int = 320) -> List[bytes]:
    """BSD-syslog lines in the shapes the reference's tests use (rfc3164_decoder.rs:215-441): 70 % standard form
    `[<pri>]Mon  D HH:MM:SS [zone] host app[pid]: text` (day padded with a space like syslogd does), 10 % with a
    leading year, 20 % custom form `[<pri>]host: YYYY Mon D HH:MM:SS [zone]: text`; 15 % carry an IANA zone name."""
    rng = np.random.default_rng(SEED_BASE + cfg)
    pool = _text_pool(rng)
    hosts = [f"h{i:04d}.dc{i % 7}.example.com" for i in range(1024)]
    apps = [f"app-{_WORDS[i % len(_WORDS)]}{i}" for i in range(64)]
    zones = ["UTC", "America/Sao_Paulo", "Europe/Paris", "Asia/Kolkata", "America/New_York", "Asia/Tokyo", "Australia/Sydney"]
    f = _ts_fields(rng, n)
    pri = rng.integers(0, 192, n)
    kind = rng.random(n)
    zk = rng.random(n)
    hi_ = rng.integers(0, 1024, n)
    ai = rng.integers(0, 64, n)
    pid = rng.integers(1, 65536, n)
    tgt = rng.integers(lo, hi + 1, n)
    po = rng.integers(0, len(pool) - 9000, n)
    invalid = rfc3164_invalid_lines()
    inv_every = int(1 / invalid_frac) if invalid_frac > 0 else 0
    out: List[bytes] = []
    for i in range(n):
        if inv_every and i % inv_every == inv_every - 1:
            out.append(invalid[(i // inv_every) % len(invalid)].encode())
            continue
        p = f"<{pri[i]}>" if pri[i] % 5 else ""
        day = f"{f['d'][i]:2d}" if kind[i] < 0.8 else str(f["d"][i])
        date = f"{_MONTHS[f['mo'][i] - 1]} {day} {f['h'][i]:02d}:{f['mi'][i]:02d}:{f['s'][i]:02d}"
        zone = " " + zones[int(zk[i] * 1000) % len(zones)] if zk[i] < 0.15 else ""
        if kind[i] < 0.7:
            head = f"{p}{date}{zone} {hosts[hi_[i]]} {apps[ai[i]]}[{pid[i]}]: "
        elif kind[i] < 0.8:
            head = f"{p}{f['y'][i]} {date}{zone} {hosts[hi_[i]]} {apps[ai[i]]}[{pid[i]}]: "
        else:
            head = f"{p}{hosts[hi_[i]]}: {f['y'][i]} {date}{zone}: {apps[ai[i]]}: "
        need = max(int(tgt[i]) - len(head), 1)
        out.append((head + pool[po[i]:po[i] + need]).encode())
    return out
