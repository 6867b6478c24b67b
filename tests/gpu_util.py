"""Helpers for the -m gpu tests: run the HIP path through the C ABI and compare with the oracle."""
from __future__ import annotations

import numpy as np

from flowgger_amd.tables import DeviceTables


def host_path_blob(dec, data, offsets):
    """fg_decode_batch (host buffers) -> canonical blob + offsets."""
    tab = dec.decode_packed(data, offsets)
    return tab.serialize(dec.fmt, data, offsets, cfg=dec._cfg), tab


def device_path(dec, data, offsets, ent_cap=None, reps=1):
    """fg_decode_batch_device on HBM-resident tensors (optionally the tile replicated `reps` times
    with rebased offsets); returns (DeviceTables, d_bytes, d_offsets)."""
    import torch

    dev = torch.device("cuda", dec.device)
    n = len(offsets) - 1
    nbytes = int(offsets[-1])
    raw = torch.from_numpy(np.ascontiguousarray(data[:nbytes])).to(dev)
    # replicas are laid out back to back WITHOUT padding so that lines stay contiguous
    d_bytes = torch.cat([raw.repeat(reps), torch.zeros(32, dtype=torch.uint8, device=dev)])
    o = torch.from_numpy(offsets[:-1].astype(np.int64)).to(dev)
    base = torch.arange(reps, device=dev, dtype=torch.int64).repeat_interleave(n) * nbytes
    d_offsets = torch.cat([o.repeat(reps) + base, torch.tensor([nbytes * reps], device=dev, dtype=torch.int64)])
    if ent_cap is None:
        ent_cap = nbytes * reps // 8 + 1024
    tables = DeviceTables(n * reps, ent_cap, dev)
    dec.decode_device(d_bytes, d_offsets, tables)
    torch.cuda.synchronize(dev)
    return tables, d_bytes, d_offsets


def first_diff(blob_a, offs_a, blob_b, offs_b, lines=None):
    """Human-readable description of the first differing line of two canonical blobs."""
    n = min(len(offs_a), len(offs_b)) - 1
    for i in range(n):
        a = blob_a[int(offs_a[i]):int(offs_a[i + 1])].tobytes()
        b = blob_b[int(offs_b[i]):int(offs_b[i + 1])].tobytes()
        if a != b:
            return f"line {i}: {lines[i] if lines is not None else ''!r}\n  gpu    {a!r}\n  oracle {b!r}"
    return f"no per-line difference (lengths {len(offs_a)} vs {len(offs_b)})"


def assert_same(blob, offs, oblob, ooffs, lines=None):
    same = np.array_equal(offs, ooffs) and np.array_equal(blob, oblob)
    assert same, first_diff(blob, offs, oblob, ooffs, lines)
