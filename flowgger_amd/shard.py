"""Multi-GPU sharding of a packed batch.

Lines decode independently, so the batch is cut into G contiguous line ranges balanced by BYTES
(fg_shard_plan), one per GPU / rank; every rank runs the same kernels on its slice and the host
concatenates the tables in shard order -- original line order is preserved without sorting and
there is NO data-path collective.  torch.distributed is only used to move the (small) tables to
rank 0 when one ordered stream is wanted ("host gather"), and by bench.py for the barrier.
Config 5 ("mixed RFC5424 + LTSV") adds `ordered_merge`: two format sub-batches are decoded
separately and re-interleaved by original line index.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

import ctypes as C

from . import _lib as L
from .tables import HostTables, _DT, layout


def shard_plan(offsets: np.ndarray, g: int) -> np.ndarray:
    """g+1 line indices: shard k owns lines [starts[k], starts[k+1])."""
    offsets = np.ascontiguousarray(offsets, np.uint64)
    starts = np.zeros(g + 1, np.uint64)
    L.check(L.lib().fg_shard_plan(offsets.ctypes.data, len(offsets) - 1, g, starts.ctypes.data), "fg_shard_plan")
    return starts.astype(np.int64)


def shard_slice(data: np.ndarray, offsets: np.ndarray, starts: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Shard k's bytes and REBASED offsets (first offset 0)."""
    a, b = int(starts[k]), int(starts[k + 1])
    lo, hi = int(offsets[a]), int(offsets[b])
    return data[lo:hi], (offsets[a:b + 1] - np.uint64(lo)).astype(np.uint64)


def _alloc_tables(n: int, ent: int) -> HostTables:
    """Host tables with room for n rows / ent entries (numpy; every array at least one element)."""
    offs, _ = layout(n, ent)
    arrays = {}
    for name, (_, size) in zip(L.TABLE_FIELDS, offs):
        dt = np.dtype(_DT[name])
        arrays[name] = np.zeros(max(size // dt.itemsize, 1), dt)
    return HostTables(n, ent, arrays)


def _part_array(parts: Sequence[HostTables]):
    arr = (L.fg_tables * max(len(parts), 1))()
    for k, p in enumerate(parts):
        C.memmove(C.byref(arr[k]), C.byref(p.struct), C.sizeof(L.fg_tables))
        arr[k].ent_cap = max(int(p.ent_cap), p.ent_used)
    return arr


def concat_tables(parts: Sequence[HostTables]) -> HostTables:
    """The ordered host gather: fg_gather_tables (C ABI) concatenates per-shard tables in shard order; entry indices
    are rebased, spans are line-relative and need no change."""
    arr = _part_array(parts)
    n, e = C.c_uint64(), C.c_uint64()
    L.check(L.lib().fg_gather_size(arr, len(parts), C.byref(n), C.byref(e)), "fg_gather_size")
    out = _alloc_tables(int(n.value), int(e.value))
    L.check(L.lib().fg_gather_tables(arr, len(parts), C.byref(out.struct)), "fg_gather_tables")
    return out


def merge_tables(parts: Sequence[HostTables], index: Sequence[np.ndarray], out: Optional[HostTables] = None,
                 src: Optional[np.ndarray] = None) -> Tuple[HostTables, np.ndarray]:
    """Config 5 on tables: sub-batches split off by format go back to their original line positions (fg_merge_tables).
    index[k][j] = original position of row j of parts[k].  Returns (tables, src_part uint8[n]).  `out` / `src`: buffers of an
    earlier call to reuse (a framer merges batch after batch into the same memory)."""
    arr = _part_array(parts)
    n, e = C.c_uint64(), C.c_uint64()
    L.check(L.lib().fg_gather_size(arr, len(parts), C.byref(n), C.byref(e)), "fg_gather_size")
    if out is None or out.n < int(n.value) or out.ent_cap < int(e.value):
        out = _alloc_tables(int(n.value), int(e.value))
    ix = [np.ascontiguousarray(i, np.uint64) for i in index]
    ptrs = (C.c_void_p * max(len(ix), 1))(*[a.ctypes.data for a in ix])
    if src is None or len(src) < int(n.value):
        src = np.zeros(max(int(n.value), 1), np.uint8)
    L.check(L.lib().fg_merge_tables(arr, len(parts), ptrs, C.byref(out.struct), src.ctypes.data), "fg_merge_tables")
    return out, src[: int(n.value)]


def merge_tables_device(dec, parts, d_index, out=None, d_src=None, stream=None):
    """fg_merge_tables_device: the same merge while the sub-batches' tables are still in HBM.  parts: DeviceTables (as the decoders
    left them); d_index[k]: torch int64 / uint64 tensor on the device, original position of row j of parts[k].  Returns
    (DeviceTables, src_part uint8 tensor); asynchronous on `stream` (default: torch's current stream).  ONE merged table then crosses the
    link (DeviceTables.to_host_pinned) instead of every part's plus a pass of the host over all of them."""
    import torch

    from .tables import DeviceTables

    dev = parts[0].buf.device
    n = sum(int(p.n) for p in parts)
    cap = sum(int(p.ent_cap) for p in parts)
    if out is None or out.n != n or out.ent_cap < cap:
        out = DeviceTables(n, cap, dev)
    if d_src is None or d_src.numel() < n:
        d_src = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
    if stream is None:
        stream = torch.cuda.current_stream(dev)
    arr = (L.fg_tables * len(parts))(*[p.struct for p in parts])
    ptrs = (C.c_void_p * len(parts))(*[int(i.data_ptr()) for i in d_index])
    L.check(L.lib().fg_merge_tables_device(dec._ctx, arr, len(parts), ptrs, C.byref(out.struct), int(d_src.data_ptr()),
                                           C.c_void_p(stream.cuda_stream)), "fg_merge_tables_device")
    return out, d_src[:n]


def decode_sharded(decode: Callable[[np.ndarray, np.ndarray, int], HostTables], data: np.ndarray,
                   offsets: np.ndarray, g: int) -> HostTables:
    """Single-process driver: decode(shard_bytes, shard_offsets, k) for each of g shards (each
    call may target GPU k), then the ordered host gather."""
    starts = shard_plan(offsets, g)
    return concat_tables([decode(*shard_slice(data, offsets, starts, k), k) for k in range(g)])


def _pack(t: HostTables) -> np.ndarray:
    """One contiguous byte buffer per shard for the wire: the table arrays in TABLE_FIELDS order, entry columns cut at
    ent_used."""
    used = t.ent_used
    offs, total = layout(t.n, used)
    buf = np.zeros(total, np.uint8)
    for name, (off, size) in zip(L.TABLE_FIELDS, offs):
        if size:
            buf[off:off + size] = t.a[name].view(np.uint8)[:size]
    return buf


def _unpack(buf: np.ndarray, n: int, used: int) -> HostTables:
    offs, _ = layout(n, used)
    arrays = {}
    for name, (off, size) in zip(L.TABLE_FIELDS, offs):
        dt = np.dtype(_DT[name])
        arrays[name] = buf[off:off + size].view(dt) if size else np.zeros(1, dt)
    return HostTables(n, used, arrays)


def decode_distributed(decode: Callable[[np.ndarray, np.ndarray], HostTables], data: np.ndarray, offsets: np.ndarray,
                       dst: int = 0, group=None) -> Optional[HostTables]:
    """One process per GPU (torch.distributed): every rank decodes its byte-balanced slice of the
    SAME packed batch; the tables travel to `dst` as ONE byte tensor per rank (sizes first, then
    point-to-point sends -- no pickling, no collective on the data path) and are put in rank order
    by fg_gather_tables.  Returns the full table on dst, None elsewhere.  Backend-agnostic
    (gloo on CPU tests; with nccl = RCCL the byte tensors are staged through the rank's GPU)."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    starts = shard_plan(offsets, world)
    mine = decode(*shard_slice(data, offsets, starts, rank))
    return gather_distributed(mine, dst, group)


def gather_distributed(mine: HostTables, dst: int = 0, group=None) -> Optional[HostTables]:
    """The ordered host gather across ranks: every rank hands in the table of ITS shard (rank order = shard order), `dst` gets
    their concatenation (fg_gather_tables), the others None.  One byte tensor per rank, sizes first, point-to-point sends -- no
    pickling, no collective on the data path."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    wire = torch.from_numpy(_pack(mine))
    dims = torch.tensor([mine.n, mine.ent_used, wire.numel()], dtype=torch.int64, device=dev)
    all_dims = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_dims, dims, group=group)
    if rank != dst:
        dist.send(wire.to(dev), dst=dst, group=group)
        return None
    parts = []
    for r in range(world):
        n_r, used_r, bytes_r = (int(x) for x in all_dims[r].tolist())
        if r == rank:
            parts.append(mine)
            continue
        buf = torch.empty(bytes_r, dtype=torch.uint8, device=dev)
        dist.recv(buf, src=r, group=group)
        parts.append(_unpack(buf.cpu().numpy(), n_r, used_r))
    return concat_tables(parts)


def ordered_merge(parts: Sequence[Tuple[np.ndarray, np.ndarray, np.ndarray]]) -> Tuple[np.ndarray, np.ndarray]:
    """Config 5's ordered gather for byte records: parts = [(orig_index[int64 m], blob uint8, offs uint64[m+1])] per
    format sub-batch (canonical Record blobs or encoded messages); returns (blob, offs) in original line order
    (fg_ordered_merge: runs of neighbouring lines move as one copy)."""
    g = len(parts)
    ix = [np.ascontiguousarray(p[0], np.uint64) for p in parts]
    blobs = [np.ascontiguousarray(p[1], np.uint8) for p in parts]
    offs = [np.ascontiguousarray(p[2], np.uint64) for p in parts]
    m = np.array([len(i) for i in ix], np.uint64)
    n = int(m.sum())
    vp = C.c_void_p
    pix = (vp * max(g, 1))(*[a.ctypes.data for a in ix])
    pbl = (vp * max(g, 1))(*[a.ctypes.data for a in blobs])
    pof = (vp * max(g, 1))(*[a.ctypes.data for a in offs])
    out_offs = np.zeros(n + 1, np.uint64)
    lib = L.lib()
    total = lib.fg_ordered_merge(g, m.ctypes.data, pix, pbl, pof, None, 0, out_offs.ctypes.data)
    if total < 0:
        raise L.FgError(int(total), "fg_ordered_merge")
    out = np.zeros(max(int(total), 1), np.uint8)
    total = lib.fg_ordered_merge(g, m.ctypes.data, pix, pbl, pof, out.ctypes.data, int(total), out_offs.ctypes.data)
    if total < 0:
        raise L.FgError(int(total), "fg_ordered_merge")
    return out[: int(total)], out_offs
