"""Record data model -- mirror of flowgger's src/flowgger/record.rs:3-82.

`Record`, `StructuredData` and `SDValue` keep the reference's field set and optionality.  A
record is materialised from one decode-table row + the line's bytes; the C ABI hands it over in
the canonical serialisation, which :func:`parse_canonical` turns into objects.
"""
from __future__ import annotations

import struct
import time
from dataclasses import dataclass, field
from typing import List, Optional, Tuple, Union

SD_STRING, SD_BOOL, SD_F64, SD_I64, SD_U64, SD_NULL = "String", "Bool", "F64", "I64", "U64", "Null"
_KINDS = [SD_STRING, SD_BOOL, SD_F64, SD_I64, SD_U64, SD_NULL]


@dataclass(frozen=True)
class SDValue:
    """record.rs:3-11 -- enum SDValue { String, Bool, F64, I64, U64, Null }"""
    kind: str
    value: Union[str, bool, float, int, None] = None

    def __repr__(self) -> str:  # Rust {:?} rendering, pinned by record.rs:95
        if self.kind == SD_NULL:
            return "Null"
        if self.kind == SD_STRING:
            return f'String("{self.value}")'
        if self.kind == SD_BOOL:
            return f"Bool({'true' if self.value else 'false'})"
        return f"{self.kind}({self.value})"


@dataclass
class StructuredData:
    """record.rs:23-27"""
    sd_id: Optional[str] = None
    pairs: List[Tuple[str, SDValue]] = field(default_factory=list)

    def __str__(self) -> str:  # impl fmt::Display, record.rs:41-68
        out = "[" + (self.sd_id or "")
        for name, value in self.pairs:
            if name.startswith("_"):
                name = name[1:]
            if value.kind == SD_NULL:
                out += f" {name}"
            elif value.kind == SD_BOOL:
                out += f' {name}="{"true" if value.value else "false"}"'
            else:
                out += f' {name}="{value.value}"'
        return out + "]"


@dataclass
class Record:
    """record.rs:70-82"""
    ts: float
    hostname: str
    facility: Optional[int] = None
    severity: Optional[int] = None
    appname: Optional[str] = None
    procid: Optional[str] = None
    msgid: Optional[str] = None
    msg: Optional[str] = None
    full_msg: Optional[str] = None
    sd: Optional[List[StructuredData]] = None


class DecodeError(Exception):
    """Err(&'static str) of Decoder::decode -- str(e) is the reference's exact message."""


def _s(b: bytes) -> str:
    return b.decode("utf-8", "surrogateescape")


def parse_canonical(buf: bytes, now: Optional[float] = None) -> Union[Record, DecodeError]:
    """Canonical serialisation (one line's worth) -> Record or DecodeError (not raised)."""
    mv = memoryview(buf)
    p = 0

    def u8():
        nonlocal p
        v = mv[p]
        p += 1
        return v

    def u32():
        nonlocal p
        (v,) = struct.unpack_from("<I", mv, p)
        p += 4
        return v

    def raw(n):
        nonlocal p
        v = bytes(mv[p:p + n])
        p += n
        return v

    def optstr():
        if u8() == 0:
            return None
        return _s(raw(u32()))

    tag = u8()
    if tag == 1:
        return DecodeError(_s(raw(u32())))
    ts_now = u8()
    (ts,) = struct.unpack("<d", raw(8))
    if ts_now:
        ts = time.time() if now is None else now  # gelf_decoder.rs:109
    fac, sev = u8(), u8()
    hostname = optstr()
    appname, procid, msgid, msg, full_msg = optstr(), optstr(), optstr(), optstr(), optstr()
    sd = None
    if u8():
        sd = []
        for _ in range(u32()):
            elem = StructuredData(optstr(), [])
            for _ in range(u32()):
                key = _s(raw(u32()))
                ty = u8()
                if ty == 0:
                    val = SDValue(SD_STRING, _s(raw(u32())))
                elif ty == 1:
                    val = SDValue(SD_BOOL, bool(u8()))
                elif ty == 2:
                    val = SDValue(SD_F64, struct.unpack("<d", raw(8))[0])
                elif ty == 3:
                    val = SDValue(SD_I64, struct.unpack("<q", raw(8))[0])
                elif ty == 4:
                    val = SDValue(SD_U64, struct.unpack("<Q", raw(8))[0])
                else:
                    val = SDValue(SD_NULL)
                elem.pairs.append((key, val))
            sd.append(elem)
    assert p == len(buf), "trailing bytes in canonical record"
    return Record(ts=ts, hostname=hostname or "", facility=None if fac == 0xFF else fac,
                  severity=None if sev == 0xFF else sev, appname=appname, procid=procid, msgid=msgid,
                  msg=msg, full_msg=full_msg, sd=sd)
