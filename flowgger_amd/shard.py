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

from . import _lib as L
from .tables import HostTables, _DT


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


def concat_tables(parts: Sequence[HostTables]) -> HostTables:
    """Concatenate per-shard tables in shard order; entry indices are rebased, spans are
    line-relative and need no change."""
    n = sum(p.n for p in parts)
    arrays = {}
    for name in L.TABLE_FIELDS:
        if name in ("ent_first", "ent_used"):
            continue
        if name.startswith("ent_") and name != "ent_count":
            chunks = [p.a[name][: p.ent_used * (2 if name == "ent_name" else 1)] for p in parts]
        elif name in ("meta", "ts", "ent_count"):
            chunks = [p.a[name][: p.n] for p in parts]
        else:  # span columns: 2 uint32 per row
            chunks = [p.a[name][: 2 * p.n] for p in parts]
        arrays[name] = np.concatenate(chunks) if chunks else np.zeros(0, _DT[name])
        if arrays[name].size == 0:
            arrays[name] = np.zeros(1, _DT[name])
    base, firsts = 0, []
    for p in parts:
        firsts.append(p.a["ent_first"][: p.n].astype(np.uint64) + np.uint64(base))
        base += p.ent_used
    arrays["ent_first"] = (np.concatenate(firsts) if firsts else np.zeros(1, np.uint64)).astype(np.uint32)
    if arrays["ent_first"].size == 0:
        arrays["ent_first"] = np.zeros(1, np.uint32)
    arrays["ent_used"] = np.array([base], np.uint64)
    return HostTables(n, base, arrays)


def decode_sharded(decode: Callable[[np.ndarray, np.ndarray, int], HostTables], data: np.ndarray,
                   offsets: np.ndarray, g: int) -> HostTables:
    """Single-process driver: decode(shard_bytes, shard_offsets, k) for each of g shards (each
    call may target GPU k), then the ordered host gather."""
    starts = shard_plan(offsets, g)
    return concat_tables([decode(*shard_slice(data, offsets, starts, k), k) for k in range(g)])


def decode_distributed(decode: Callable[[np.ndarray, np.ndarray], HostTables], data: np.ndarray, offsets: np.ndarray,
                       dst: int = 0, group=None) -> Optional[HostTables]:
    """One process per GPU (torch.distributed): every rank decodes its byte-balanced slice of the
    SAME packed batch; tables are gathered to `dst` in rank order.  Returns the full table on dst,
    None elsewhere.  Backend-agnostic (gloo on CPU tests, nccl = RCCL on GPUs)."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    starts = shard_plan(offsets, world)
    mine = decode(*shard_slice(data, offsets, starts, rank))
    payload = {"n": mine.n, "used": mine.ent_used, "a": mine.a}
    gathered: Optional[List] = [None] * world if rank == dst else None
    dist.gather_object(payload, gathered, dst=dst, group=group)
    if rank != dst:
        return None
    return concat_tables([HostTables(p["n"], p["used"], p["a"]) for p in gathered])


def ordered_merge(parts: Sequence[Tuple[np.ndarray, np.ndarray, np.ndarray]]) -> Tuple[np.ndarray, np.ndarray]:
    """Config 5's ordered gather: parts = [(orig_index[int64 m], blob uint8, offs uint64[m+1])] per
    format sub-batch (canonical Record blobs); returns (blob, offs) in original line order."""
    n = sum(len(ix) for ix, _, _ in parts)
    sizes = np.zeros(n, np.int64)
    for ix, _, offs in parts:
        sizes[ix] = np.diff(offs.astype(np.int64))
    out_offs = np.zeros(n + 1, np.uint64)
    out_offs[1:] = np.cumsum(sizes)
    out = np.zeros(int(out_offs[-1]), np.uint8)
    for ix, blob, offs in parts:
        o = offs.astype(np.int64)
        for j, i in enumerate(ix):  # per-line copy; sub-batches are already in relative order
            out[int(out_offs[i]):int(out_offs[i + 1])] = blob[o[j]:o[j + 1]]
    return out, out_offs
