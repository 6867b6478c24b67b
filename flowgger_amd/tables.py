"""Decode tables (include/fg_hip.h `fg_tables`) as numpy (host) or torch (device) memory.

PyTorch is used only as the device allocator / stream provider; every compute call goes through
the C ABI with raw pointers.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib as L

_DT = {
    "meta": np.uint32, "ts": np.float64, "hostname": np.uint32, "appname": np.uint32, "procid": np.uint32,
    "msgid": np.uint32, "msg": np.uint32, "full_msg": np.uint32, "ent_first": np.uint32,
    "ent_count": np.uint32, "ent_name": np.uint32, "ent_val": np.uint64, "ent_type": np.uint8,
    "ent_flags": np.uint8, "ent_used": np.uint64,
}


def layout(n: int, ent_cap: int) -> Tuple[list, int]:
    """Byte sizes of the 15 arrays and 256-byte-aligned offsets inside one allocation."""
    sizes = (C.c_uint64 * L.FG_TABLE_ARRAYS)()
    L.check(L.lib().fg_tables_layout(n, ent_cap, sizes), "fg_tables_layout")
    offs, off = [], 0
    for s in sizes:
        offs.append((off, int(s)))
        off += (int(s) + 255) // 256 * 256
    return offs, max(off, 256)


class HostTables:
    """Tables in host memory (numpy).  Built from a device copy or from fg_decode_batch output."""

    def __init__(self, n: int, ent_cap: int, arrays: dict):
        self.n, self.ent_cap, self.a = n, ent_cap, arrays
        self.struct = L.fg_tables()
        self.struct.n, self.struct.ent_cap = n, ent_cap
        for name in L.TABLE_FIELDS:
            setattr(self.struct, name, arrays[name].ctypes.data)

    @classmethod
    def from_struct(cls, st: "L.fg_tables") -> "HostTables":
        """Copy out of ctx-owned pinned memory (valid only until the next call on that ctx)."""
        n = int(st.n)
        used = int(np.ctypeslib.as_array(C.cast(st.ent_used, C.POINTER(C.c_uint64)), (1,))[0])
        offs, _ = layout(n, used)
        arrays = {}
        for name, (_, size) in zip(L.TABLE_FIELDS, offs):
            dt = np.dtype(_DT[name])
            cnt = size // dt.itemsize
            ptr = getattr(st, name)
            if cnt == 0 or not ptr:
                arrays[name] = np.zeros(max(cnt, 1), dt)
            else:
                arrays[name] = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (size,)).view(dt).copy()
        return cls(n, used, arrays)

    # convenience views -------------------------------------------------------------------
    @property
    def status(self) -> np.ndarray:
        return (self.a["meta"] & 0xFF).astype(np.uint8)

    @property
    def ent_used(self) -> int:
        return int(self.a["ent_used"][0])

    def span(self, name: str) -> np.ndarray:
        return self.a[name].reshape(-1, 2)

    def serialize(self, fmt: int, bytes_np: np.ndarray, offsets_np: np.ndarray, i0: int = 0,
                  i1: Optional[int] = None, cfg=None) -> Tuple[np.ndarray, np.ndarray]:
        """Canonical Record serialisation of rows [i0, i1): (blob uint8, offsets uint64[i1-i0+1])."""
        i1 = self.n if i1 is None else i1
        lib = L.lib()
        offs = np.zeros(i1 - i0 + 1, np.uint64)
        cfgp = C.byref(cfg) if cfg is not None else None
        total = lib.fg_tables_serialize(fmt, cfgp, bytes_np.ctypes.data, offsets_np.ctypes.data,
                                        C.byref(self.struct), i0, i1, None, 0, offs.ctypes.data)
        if total < 0:
            raise L.FgError(int(total), "fg_tables_serialize")
        blob = np.zeros(max(int(total), 1), np.uint8)
        lib.fg_tables_serialize(fmt, cfgp, bytes_np.ctypes.data, offsets_np.ctypes.data,
                                C.byref(self.struct), i0, i1, blob.ctypes.data, int(total), offs.ctypes.data)
        return blob[:int(total)], offs


def tables_stdout(t: HostTables, fmt: int, bytes_np: np.ndarray, offsets_np: np.ndarray, framing: int = 0) -> bytes:
    """fg_tables_stdout: what the reference decoders print to stdout while decoding these rows (ltsv_decoder.rs:99)."""
    lib = L.lib()
    args = (fmt, framing, bytes_np.ctypes.data, offsets_np.ctypes.data, C.byref(t.struct), 0, t.n)
    n = lib.fg_tables_stdout(*args, None, 0)
    if n < 0:
        raise L.FgError(int(n), "fg_tables_stdout")
    buf = np.zeros(max(int(n), 1), np.uint8)
    lib.fg_tables_stdout(*args, buf.ctypes.data, int(n))
    return buf[: int(n)].tobytes()


class DeviceTables:
    """Tables in HBM: one torch uint8 allocation carved into the 15 arrays."""

    def __init__(self, n: int, ent_cap: int, device="cuda:0"):
        import torch

        self.n, self.ent_cap = n, ent_cap
        self.offs, total = layout(n, ent_cap)
        self.buf = torch.empty(total, dtype=torch.uint8, device=device)
        base = self.buf.data_ptr()
        assert base % 256 == 0
        self.struct = L.fg_tables()
        self.struct.n, self.struct.ent_cap = n, ent_cap
        for name, (off, _) in zip(L.TABLE_FIELDS, self.offs):
            setattr(self.struct, name, base + off)

    def column(self, name: str):
        """torch uint8 view of one array (device)."""
        off, size = self.offs[L.TABLE_FIELDS.index(name)]
        return self.buf[off:off + size]

    def to_host(self, allow_overflow: bool = False) -> HostTables:
        """Copy to the host.  allow_overflow: a table whose entries did not all fit (rows with status FG_ST_OVERFLOW) is
        returned as it is -- every other row is valid (include/fg_hip.h) -- instead of raising."""
        import torch

        used = int(self.column("ent_used").view(torch.int64)[0].item())
        if used > self.ent_cap:
            if not allow_overflow:
                raise L.FgError(L.FG_ERR_ENT_OVERFLOW, f"entry table overflow ({used} > {self.ent_cap})")
            used = self.ent_cap
        arrays = {}
        for name, (off, size) in zip(L.TABLE_FIELDS, self.offs):
            dt = np.dtype(_DT[name])
            if name.startswith("ent_") and name not in ("ent_first", "ent_count", "ent_used"):
                size = used * (size // self.ent_cap if self.ent_cap else 0)
            if size == 0:
                arrays[name] = np.zeros(1, dt)
            else:
                arrays[name] = self.buf[off:off + size].cpu().numpy().view(dt).copy()
        return HostTables(self.n, used, arrays)

    def to_host_pinned(self, stream=None, host=None) -> HostTables:
        """The D2H leg of the host gather as a framer would run it: every column (entry columns up to ent_used) into ONE pinned
        host buffer with asynchronous copies on `stream`, one synchronisation at the end.  The HostTables views that buffer."""
        import torch

        used = int(self.column("ent_used").view(torch.int64)[0].item())
        if used > self.ent_cap:
            raise L.FgError(L.FG_ERR_ENT_OVERFLOW, f"entry table overflow ({used} > {self.ent_cap})")
        offs, total = layout(self.n, used)
        if host is None or host.numel() < total:  # (`host`: the pinned buffer of an earlier call, HostTables._pinned)
            host = torch.empty(total, dtype=torch.uint8, pin_memory=True)
        ctx = torch.cuda.stream(stream) if stream is not None else None
        if ctx is not None:
            ctx.__enter__()
        try:
            for name, (doff, _), (hoff, size) in zip(L.TABLE_FIELDS, self.offs, offs):
                if size:
                    host[hoff:hoff + size].copy_(self.buf[doff:doff + size], non_blocking=True)
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
        (stream if stream is not None else torch.cuda.current_stream(self.buf.device)).synchronize()
        flat = host.numpy()
        arrays = {}
        for name, (hoff, size) in zip(L.TABLE_FIELDS, offs):
            dt = np.dtype(_DT[name])
            arrays[name] = flat[hoff:hoff + size].view(dt) if size else np.zeros(1, dt)
        t = HostTables(self.n, used, arrays)
        t._pinned = host  # keeps the buffer alive
        return t
