"""The IANA time-zone table the RFC3164 decoder needs (reference: rfc3164_decoder.rs:193-203 --
``time_tz::timezones::get_by_name(token)`` + ``PrimitiveDateTime::assume_timezone``; the time-tz crate embeds the
tz database at build time).  Here the table is built on the host from the system's compiled TZif files (the `tzdata`
Python package or /usr/share/zoneinfo) and handed to the GPU decoder through the C ABI (fg_rfc3164_cfg):

    names      sorted zone names ("UTC", "America/Sao_Paulo", ...) -- exact, case-sensitive match
    per zone   (utc_start, utc_offset) entries: the offset in effect from utc_start on; the first entry starts at
               -2^63.  Explicit TZif (v2+, 64-bit) transitions, then the footer's POSIX TZ rule (Mm.w.d / Jn / n forms)
               expanded year by year up to `until_year`.

Product code (configuration data, like the LTSV schema); no decode logic lives here.
"""
from __future__ import annotations

import calendar
import os
import re
import struct
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

I64_MIN = -(2 ** 63)


def _tz_root_candidates():
    roots = []
    try:
        import importlib.resources as ir

        roots.append(str(ir.files("tzdata") / "zoneinfo"))
    except Exception:
        pass
    roots += ["/usr/share/zoneinfo", "/usr/lib/zoneinfo", "/usr/share/lib/zoneinfo", "/etc/zoneinfo"]
    return [r for r in roots if os.path.isdir(r)]


def available_zone_files() -> Dict[str, str]:
    """zone name -> TZif path (first root that has the name wins)."""
    out: Dict[str, str] = {}
    for root in _tz_root_candidates():
        for d, _, files in os.walk(root):
            for f in files:
                p = os.path.join(d, f)
                name = os.path.relpath(p, root)
                if name in out or name.startswith(("posix/", "right/")) or "." in f or f in ("leapseconds", "tzdata.zi", "zone.tab",
                                                                                             "zone1970.tab", "iso3166.tab", "+VERSION"):
                    continue
                try:
                    with open(p, "rb") as fh:
                        if fh.read(4) != b"TZif":
                            continue
                except OSError:
                    continue
                out[name.replace(os.sep, "/")] = p
    return out


def _parse_tzif(data: bytes) -> Tuple[List[int], List[int], List[int], str]:
    """-> (transition times, type index per transition, utoff per type, footer)"""
    def block(off, tsize):
        magic, ver, isutc, isstd, leap, timecnt, typecnt, charcnt = struct.unpack(">4sc15x6I", data[off:off + 44])
        assert magic == b"TZif"
        p = off + 44
        fmt = ">%d%s" % (timecnt, "q" if tsize == 8 else "i")
        times = list(struct.unpack(fmt, data[p:p + timecnt * tsize]))
        p += timecnt * tsize
        idx = list(data[p:p + timecnt])
        p += timecnt
        utoff = []
        for k in range(typecnt):
            o, _dst, _ab = struct.unpack(">iBB", data[p + 6 * k:p + 6 * k + 6])
            utoff.append(o)
        p += 6 * typecnt + charcnt + leap * (tsize + 4) + isstd + isutc
        return ver, times, idx, utoff, p

    ver, times, idx, utoff, end = block(0, 4)
    footer = ""
    if ver >= b"2":
        ver, times, idx, utoff, end = block(end, 8)
        nl = data.find(b"\n", end + 1)
        footer = data[end + 1:nl].decode("ascii", "replace") if data[end:end + 1] == b"\n" and nl > 0 else ""
    return times, idx, utoff, footer


_OFF = r"([+-]?\d{1,3})(?::(\d{1,2}))?(?::(\d{1,2}))?"
_NAME = r"(?:<[^>]+>|[A-Za-z]{3,})"


def _secs(h, m, s, neg_is_east=True):
    sign = -1 if h.startswith("-") else 1
    v = abs(int(h)) * 3600 + int(m or 0) * 60 + int(s or 0)
    return sign * v


def _rule_day_of_year_secs(rule: str, year: int) -> int:
    """seconds since the start of `year` (local wall clock of the rule) at which the rule fires"""
    m = re.fullmatch(r"(M(\d+)\.(\d+)\.(\d+)|J(\d+)|(\d+))(?:/" + _OFF + ")?", rule)
    if not m:
        raise ValueError(rule)
    t = _secs(m.group(7), m.group(8), m.group(9)) if m.group(7) is not None else 7200
    if m.group(2) is not None:
        mon, week, wd = int(m.group(2)), int(m.group(3)), int(m.group(4))
        first_wd = (calendar.weekday(year, mon, 1) + 1) % 7  # 0 = Sunday
        day = 1 + (wd - first_wd) % 7 + 7 * (week - 1)
        while day > calendar.monthrange(year, mon)[1]:
            day -= 7
        doy = (calendar.timegm((year, mon, day, 0, 0, 0)) - calendar.timegm((year, 1, 1, 0, 0, 0))) // 86400
    elif m.group(5) is not None:  # Jn: 1..365, Feb 29 never counted
        n = int(m.group(5))
        doy = n - 1 + (1 if calendar.isleap(year) and n >= 60 else 0)
    else:  # n: 0..365, leap days counted
        doy = int(m.group(6))
    return doy * 86400 + t


def _expand_footer(footer: str, after_utc: int, until_year: int) -> List[Tuple[int, int]]:
    """(utc_start, utoff) entries produced by the POSIX TZ string for instants > after_utc"""
    m = re.fullmatch(_NAME + _OFF + r"(?:(" + _NAME + r")(?:" + _OFF + r")?,([^,]+),([^,]+))?", footer)
    if not m:
        return []
    std = -_secs(m.group(1), m.group(2), m.group(3))
    if m.group(4) is None:
        return [(after_utc, std)] if after_utc == I64_MIN else []
    dst = -_secs(m.group(5), m.group(6), m.group(7)) if m.group(5) is not None else std + 3600
    start_rule, end_rule = m.group(8), m.group(9)
    import time as _t

    y0 = 1970 if after_utc == I64_MIN else max(1900, _t.gmtime(max(after_utc, -2208988800)).tm_year - 1)
    out = []
    for y in range(y0, until_year + 1):
        jan1 = calendar.timegm((y, 1, 1, 0, 0, 0))
        s = jan1 + _rule_day_of_year_secs(start_rule, y) - std   # DST starts (wall clock = standard time)
        e = jan1 + _rule_day_of_year_secs(end_rule, y) - dst     # DST ends (wall clock = DST)
        for tt, off in sorted([(s, dst), (e, std)]):
            if tt > after_utc:
                out.append((tt, off))
    return out


@dataclass
class TzTable:
    names: List[str]
    zone_first: np.ndarray   # uint32[nz + 1] into the entry arrays
    utc_start: np.ndarray    # int64[total]
    utc_offset: np.ndarray   # int32[total]

    def entries(self, name: str):
        k = self.names.index(name)
        a, b = int(self.zone_first[k]), int(self.zone_first[k + 1])
        return self.utc_start[a:b], self.utc_offset[a:b]


def zone_entries(path: str, until_year: int = 2100) -> List[Tuple[int, int]]:
    with open(path, "rb") as fh:
        times, idx, utoff, footer = _parse_tzif(fh.read())
    # the offset before the first transition: the first non-DST type, in practice type 0 (RFC 8536 3.2)
    ents: List[Tuple[int, int]] = [(I64_MIN, utoff[0] if utoff else 0)]
    for t, k in zip(times, idx):
        if ents and ents[-1][0] == t:
            ents[-1] = (t, utoff[k])
        else:
            ents.append((t, utoff[k]))
    if footer:
        ents += _expand_footer(footer, ents[-1][0] if times else I64_MIN, until_year) if times else \
            [(t, o) for t, o in _expand_footer(footer, I64_MIN, until_year) if t != I64_MIN]
        if not times:
            m = _expand_footer(footer, I64_MIN, 1969)  # fixed-offset footer: take its offset as the initial one
            if m and m[0][0] == I64_MIN:
                ents[0] = (I64_MIN, m[0][1])
    # drop no-op entries
    out = [ents[0]]
    for t, o in ents[1:]:
        if o != out[-1][1]:
            out.append((t, o))
    return out


def build_table(zones: Optional[Iterable[str]] = None, until_year: int = 2100) -> TzTable:
    files = available_zone_files()
    names = sorted(files if zones is None else [z for z in zones if z in files], key=lambda s: s.encode())
    first, starts, offs = [0], [], []
    for nm in names:
        ents = zone_entries(files[nm], until_year)
        starts += [t for t, _ in ents]
        offs += [o for _, o in ents]
        first.append(len(starts))
    return TzTable(names, np.array(first, np.uint32), np.array(starts, np.int64), np.array(offs, np.int32))


_default: Optional[TzTable] = None


def default_table() -> TzTable:
    global _default
    if _default is None:
        _default = build_table()
    return _default
