#!/usr/bin/env python3
"""Probe (GPU box): do the decode kernels run at LINK speed when they read their input straight out of pinned host memory and write their
tables straight into pinned host memory -- no hipMemcpy, no slices, no events -- and are both directions of the link busy at once then?
(profiles/r04a_timeline_*: with hipMemcpyAsync on two streams the runtime put uploads and downloads on ONE copy queue: an upload and a
download were never in flight together.)

For each corpus: fg_decode_batch_device called with
  in=host,out=host   bytes + offsets in pinned host memory, table columns in pinned host memory (ent_used stays in HBM: atomics)
  in=host,out=hbm    tables in HBM (what the input side alone gives)
  in=hbm,out=host    input resident in HBM (what the output side alone gives)
and fg_decode_batch (the sliced hipMemcpy pipeline of round 3) beside them.  Every variant's rows are compared with the HBM -> HBM result.
usage: python tools/probe/zero_copy.py [corpora: cfg2,cfg3,ltsv,cfg4] [lines per tile = 250000] [reps = 8]
"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from flowgger_amd import GelfDecoder, LTSVDecoder, RFC5424Decoder, synth  # noqa: E402
from flowgger_amd import _lib as L  # noqa: E402
from flowgger_amd.tables import layout  # noqa: E402

dev = torch.device("cuda", 0)
lib = L.lib()


def pinned(nbytes):
    p = C.c_void_p()
    L.check(lib.fg_alloc_pinned(nbytes, C.byref(p)), "fg_alloc_pinned")
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (nbytes,)), p


def carve(base_ptr, n, ent_cap, ent_used_ptr):
    offs, total = layout(n, ent_cap)
    st = L.fg_tables()
    st.n, st.ent_cap = n, ent_cap
    for name, (off, _) in zip(L.TABLE_FIELDS, offs):
        setattr(st, name, base_ptr + off)
    st.ent_used = ent_used_ptr
    return st, offs, total


def main():
    corpora = (sys.argv[1] if len(sys.argv) > 1 else "cfg2,cfg3,ltsv,cfg4").split(",")
    n_tile = int(sys.argv[2]) if len(sys.argv) > 2 else 250_000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    res = {}
    for name in corpora:
        if name == "cfg3":
            dec, lines = GelfDecoder(), synth.gelf_lines(n_tile)
        elif name == "ltsv":
            dec, lines = LTSVDecoder(synth.LTSV_CONFIG), synth.ltsv_lines(n_tile)
        elif name == "cfg4":
            dec, lines = RFC5424Decoder(), synth.rfc5424_lines(n_tile, cfg=4, sd=True)
        else:
            dec, lines = RFC5424Decoder(), synth.rfc5424_lines(n_tile, cfg=2)
        data, offsets = synth.pack(lines)
        tb = int(offsets[-1])
        n, nbytes = n_tile * reps, tb * reps
        h_bytes, _hb = pinned(nbytes + 64)
        h_offs8, _ho = pinned((n + 1) * 8)
        h_offs = h_offs8.view(np.uint64)
        for r in range(reps):
            h_bytes[r * tb:(r + 1) * tb] = data[:tb]
            h_offs[r * n_tile:(r + 1) * n_tile] = offsets[:-1] + np.uint64(r * tb)
        h_offs[n] = nbytes
        h_bytes[nbytes:] = 0
        d_bytes = torch.from_numpy(h_bytes).to(dev)
        d_offs = torch.from_numpy(h_offs.view(np.int64)).to(dev)
        ent_cap = nbytes // 16 + (1 << 20)
        _, _, total = carve(0, n, ent_cap, 0)
        d_tab = torch.zeros(total, dtype=torch.uint8, device=dev)
        d_used = torch.zeros(8, dtype=torch.uint8, device=dev)
        h_tab, _ht = pinned(total)
        st_dev, offs_l, _ = carve(d_tab.data_ptr(), n, ent_cap, d_used.data_ptr())
        st_host, _, _ = carve(h_tab.ctypes.data, n, ent_cap, d_used.data_ptr())
        stream = torch.cuda.current_stream(dev)

        def run(bptr, optr, st, iters=3):
            ms = []
            for it in range(iters + 1):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                L.check(lib.fg_decode_batch_device(dec._ctx, dec.fmt, bptr, nbytes, optr, n, C.byref(st), stream.cuda_stream), "decode")
                b.record(stream)
                torch.cuda.synchronize(dev)
                if it:
                    ms.append(a.elapsed_time(b))
            return min(ms)

        def rows(buf):  # the fixed columns (76 B per line) as one byte string
            return b"".join(bytes(buf[o:o + s]) for (o, s), f in zip(offs_l, L.TABLE_FIELDS) if f in
                            ("meta", "ts", "hostname", "appname", "procid", "msgid", "msg", "full_msg", "ent_count"))

        out = {"lines": n, "bytes_in": nbytes}
        t_ref = run(d_bytes.data_ptr(), d_offs.data_ptr(), st_dev)
        ref = rows(d_tab.cpu().numpy())
        used = int(d_used.cpu().numpy().view(np.uint64)[0])
        out_bytes = n * 76 + used * 18
        out["bytes_out"] = out_bytes
        out["hbm_to_hbm_ms"] = t_ref
        for tag, bptr, optr, st, buf in (("in_host_out_host", h_bytes.ctypes.data, h_offs.ctypes.data, st_host, "h"),
                                         ("in_host_out_hbm", h_bytes.ctypes.data, h_offs.ctypes.data, st_dev, "d"),
                                         ("in_hbm_out_host", d_bytes.data_ptr(), d_offs.data_ptr(), st_host, "h")):
            try:
                if buf == "h":
                    h_tab[:] = 0
                else:
                    d_tab.zero_()
                ms = run(bptr, optr, st)
                got = rows(h_tab if buf == "h" else d_tab.cpu().numpy())
                out[tag] = {"ms": ms, "M_lines_per_s": n / ms / 1e3, "GBps_in": nbytes / ms / 1e6, "GBps_out": out_bytes / ms / 1e6,
                            "rows_equal_hbm_result": got == ref}
            except Exception as e:  # noqa: BLE001
                out[tag] = {"error": repr(e)[:200]}
        # the round-3 pipeline beside it
        st2 = L.fg_tables()
        L.check(lib.fg_decode_batch(dec._ctx, dec.fmt, h_bytes.ctypes.data, nbytes, h_offs.ctypes.data, n, C.byref(st2)), "fg_decode_batch")
        t0 = time.perf_counter()
        for _ in range(3):
            L.check(lib.fg_decode_batch(dec._ctx, dec.fmt, h_bytes.ctypes.data, nbytes, h_offs.ctypes.data, n, C.byref(st2)), "fg_decode_batch")
        dt = (time.perf_counter() - t0) / 3
        out["fg_decode_batch_sliced_memcpy"] = {"ms": dt * 1e3, "M_lines_per_s": n / dt / 1e6, "GBps_in": nbytes / dt / 1e9}
        res[name] = out
        print(name, json.dumps(out), flush=True)
        for h in (_hb, _ho, _ht):
            lib.fg_free_pinned(h)
        del d_bytes, d_offs, d_tab
        torch.cuda.empty_cache()
    gb = (C.c_double * 3)()
    if lib.fg_measure_link(dec._ctx, 1 << 30, gb) == 0:
        res["link_peak_GBps"] = {"h2d": gb[0], "d2h": gb[1], "bidir_sum": gb[2]}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
