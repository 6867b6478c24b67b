"""CPU: the N>1 path -- byte-balanced sharding, ordered host gather (single process and
world_size-2 torch.distributed over gloo), and config 5's ordered merge of format sub-batches.
A deterministic numpy stand-in produces table rows, so the plumbing is tested without a GPU."""
import os

import numpy as np
import pytest

from flowgger_amd import _lib as L
from flowgger_amd import shard, synth
from flowgger_amd.tables import HostTables, _DT


def fake_decode(data, offsets, k=0):
    """Content-derived table rows (shard-independent by construction)."""
    offsets = offsets.astype(np.int64)
    n = len(offsets) - 1
    lens = np.diff(offsets)
    first = np.array([data[offsets[i]] if lens[i] else 0 for i in range(n)], np.int64)
    cnt = (lens % 3).astype(np.uint32)
    used = int(cnt.sum())
    a = {name: np.zeros(1, _DT[name]) for name in L.TABLE_FIELDS}
    a["meta"] = ((first << 8) | (lens % 251)).astype(np.uint32)
    a["ts"] = lens.astype(np.float64) * 0.5
    for j, col in enumerate(("hostname", "appname", "procid", "msgid", "msg", "full_msg")):
        a[col] = np.stack([np.full(n, j, np.uint32), lens.astype(np.uint32)], 1).reshape(-1)
    a["ent_count"] = cnt
    a["ent_first"] = (np.cumsum(cnt) - cnt).astype(np.uint32)
    vals, names = [], []
    for i in range(n):
        for e in range(int(cnt[i])):
            vals.append(int(first[i]) * 1000 + e)
            names.append((e, int(lens[i])))
    a["ent_val"] = np.array(vals or [0], np.uint64)
    a["ent_name"] = np.array(names or [(0, 0)], np.uint32).reshape(-1)
    a["ent_type"] = np.zeros(max(used, 1), np.uint8)
    a["ent_flags"] = np.zeros(max(used, 1), np.uint8)
    a["ent_used"] = np.array([used], np.uint64)
    return HostTables(n, used, a)


def rows(t: HostTables):
    out = []
    sp = {c: t.a[c].reshape(-1, 2) for c in ("hostname", "appname", "procid", "msgid", "msg", "full_msg", "ent_name")}
    for i in range(t.n):
        f, c = int(t.a["ent_first"][i]), int(t.a["ent_count"][i])
        out.append((int(t.a["meta"][i]), float(t.a["ts"][i]), tuple(tuple(sp[k][i]) for k in list(sp)[:6]),
                    tuple(int(v) for v in t.a["ent_val"][f:f + c]), tuple(tuple(x) for x in sp["ent_name"][f:f + c])))
    return out


@pytest.fixture(scope="module")
def corpus():
    lines = synth.rfc5424_lines(3000, cfg=5, sd=True, long_tail=True)
    return synth.pack(lines)


@pytest.mark.parametrize("g", [1, 2, 3, 8])
def test_sharded_equals_unsharded(corpus, g):
    data, offsets = corpus
    full = fake_decode(data, offsets)
    got = shard.decode_sharded(fake_decode, data, offsets, g)
    assert got.n == full.n and got.ent_used == full.ent_used
    assert rows(got) == rows(full)
    starts = shard.shard_plan(offsets, g)
    per = np.diff(offsets[starts].astype(np.int64))
    assert per.sum() == int(offsets[-1]) and per.max() - per.min() <= 2 * 8192


def _worker(rank, world, port, tmp):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lines = synth.rfc5424_lines(3000, cfg=5, sd=True, long_tail=True)
        data, offsets = synth.pack(lines)
        got = shard.decode_distributed(fake_decode, data, offsets, dst=0)
        if rank == 0:
            assert rows(got) == rows(fake_decode(data, offsets))
            open(os.path.join(tmp, "ok"), "w").write("1")
        else:
            assert got is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_distributed_gather_gloo_world2(tmp_path):
    import torch.multiprocessing as mp

    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_ordered_merge_of_format_sub_batches(oracle):
    """Config 5: RFC5424 and LTSV lines interleaved; each sub-batch is decoded on its own and the
    results are re-interleaved by original index."""
    a = synth.rfc5424_lines(300, cfg=5, long_tail=True)
    b = synth.ltsv_lines(200, long_tail=True)
    rng = np.random.default_rng(5)
    tag = rng.permutation(np.array([0] * len(a) + [1] * len(b)))
    ia, ib = np.nonzero(tag == 0)[0], np.nonzero(tag == 1)[0]
    da, oa = synth.pack(a)
    db, ob = synth.pack(b)
    ba, offa = oracle.decode_batch(0, da, oa)
    bb, offb = oracle.decode_batch(1, db, ob, synth.LTSV_CONFIG)
    blob, offs = shard.ordered_merge([(ia, ba, offa), (ib, bb, offb)])
    it = {0: iter(a), 1: iter(b)}
    for i, t in enumerate(tag):
        line = next(it[int(t)])
        want = oracle.decode(int(t), line, synth.LTSV_CONFIG if t else None)
        assert blob[int(offs[i]):int(offs[i + 1])].tobytes() == want
