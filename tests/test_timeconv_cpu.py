"""CPU: the host build of the kernels' timestamp arithmetic (fg_timeconv.hpp).

The reference computes ts = (unix_timestamp_nanos as f64) / 1e9 (utils/mod.rs:23-28).  The kernels
replace the IEEE division by a proven-exact 3-operation sequence (div_by_1e9) and use a
branch-free calendar conversion on the common path; both are checked here bit for bit against
the hardware divider / Python big-integer arithmetic, and the branch-free form against the
general one on every field combination that matters."""
import ctypes as C
import datetime as dt
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "native" / "timeconv_host.cpp"
LIB = ROOT / "tests" / "native" / "libtimeconv_host.so"


@pytest.fixture(scope="module")
def tc():
    hdrs = [ROOT / "flowgger_amd/csrc/fg_timeconv.hpp", SRC]
    if not LIB.exists() or LIB.stat().st_mtime < max(h.stat().st_mtime for h in hdrs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math", "-ffp-contract=off",
                        "-mfma", "-o", str(LIB), str(SRC)], check=True)
    L = C.CDLL(str(LIB))
    L.fgt_div1e9_sweep.restype = C.c_uint64
    L.fgt_div1e9_sweep.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
    return L


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5, 6, 7])
def test_div_by_1e9_equals_ieee_division(tc, mode):
    bad = C.c_double()
    n = tc.fgt_div1e9_sweep(0x5424 + mode, 12_500_000, mode, C.byref(bad))
    assert n == 0, f"{n} mismatches, first x = {bad.value!r}"


def test_div_by_1e9_structured_values(tc):
    # powers of two, neighbours of multiples of 1e9, the largest u64s, zero
    xs = [0.0, 1.0, 2.0 ** 63, 2.0 ** 64 - 2048, 1e9, 1e9 - 1, 1e9 + 1, 5e8, 1.5e9]
    xs += [float(k * 10 ** 9 + d) for k in (1, 7, 1438790025, 2 ** 33, 17179869183) for d in (-1, 0, 1, 499999999, 500000000, 500000001)]
    xs += [float(2 ** k) for k in range(0, 64)] + [float(2 ** k - 1) for k in range(1, 54)] + [float(2 ** k + 1) for k in range(1, 53)]
    x = np.array(xs + [-v for v in xs if v != 0.0], dtype=np.float64)  # (-0.0 never occurs: a negative total is nonzero)
    out = np.empty_like(x)
    tc.fgt_div1e9_batch(_p(x, C.c_double), _p(out, C.c_double), C.c_uint64(len(x)))
    ref = x / np.float64(1e9)
    bad = np.flatnonzero(out.view(np.uint64) != ref.view(np.uint64))
    assert len(bad) == 0, [(x[i], out[i], ref[i]) for i in bad[:5]]


def test_unix_nanos_to_f64_matches_big_integer_arithmetic(tc):
    rng = np.random.default_rng(7)
    secs = np.concatenate([
        rng.integers(-2 ** 38, 2 ** 38, 200_000), rng.integers(0, 2 ** 34, 200_000), rng.integers(-377705116800, 253402300800, 200_000),
        np.array([0, -1, 1, 2 ** 34 - 1, 2 ** 34, 18446744073, 18446744074, -18446744073, -18446744074, 1438790025])]).astype(np.int64)
    nano = rng.integers(0, 10 ** 9, len(secs)).astype(np.uint32)
    nano[-10:] = [0, 1, 999999999, 709551615, 709551616, 709551615, 709551616, 709551615, 709551616, 637824000]
    out = np.empty(len(secs), dtype=np.float64)
    tc.fgt_unix_nanos_batch(_p(secs, C.c_int64), _p(nano, C.c_uint32), _p(out, C.c_double), C.c_uint64(len(secs)))
    # Python: int -> float is correctly rounded (RNE), float / float is IEEE division
    idx = np.concatenate([np.arange(0, len(secs), 37), np.arange(len(secs) - 10, len(secs))])
    for i in idx:
        ref = float(int(secs[i]) * 10 ** 9 + int(nano[i])) / 1e9
        assert out[i] == ref, (int(secs[i]), int(nano[i]), out[i], ref)
    assert out[-1] == 1438790025.637824  # rfc5424_decoder.rs:246 known answer


def test_fast_calendar_path_agrees_with_general_path(tc):
    rng = np.random.default_rng(11)
    n = 400_000
    parts = np.stack([
        rng.choice([0, 1, 4, 100, 400, 1600, 1969, 1970, 1971, 2000, 2015, 2024, 2038, 2100, 2513, 2514, 2515, 9999], n),
        rng.integers(0, 14, n), rng.integers(0, 33, n), rng.integers(0, 25, n), rng.integers(0, 61, n), rng.integers(0, 62, n),
        rng.integers(0, 10 ** 9, n), rng.choice([-1, 1], n), rng.integers(0, 27, n), rng.integers(0, 61, n)], axis=1).astype(np.int32)
    # half of the rows: plausible log stamps
    m = n // 2
    parts[:m, 0] = rng.integers(1969, 2040, m)
    parts[:m, 1] = rng.integers(1, 13, m)
    parts[:m, 2] = rng.integers(1, 32, m)
    parts[:m, 3] = rng.integers(0, 24, m)
    parts[:m, 4] = rng.integers(0, 60, m)
    parts[:m, 5] = rng.integers(0, 60, m)
    parts[:m, 8] = rng.integers(0, 15, m)
    parts[:m, 9] = rng.choice([0, 15, 30, 45], m)
    parts = np.ascontiguousarray(parts)
    rc = np.empty(n, dtype=np.int32)
    rcf = np.empty(n, dtype=np.int32)
    out = np.empty(n, dtype=np.float64)
    outf = np.empty(n, dtype=np.float64)
    tc.fgt_datetime_batch(_p(parts, C.c_int32), C.c_uint64(n), 1, _p(rc, C.c_int32), _p(out, C.c_double), _p(rcf, C.c_int32), _p(outf, C.c_double))
    decided = rcf != 2
    assert decided[:m].mean() > 0.95                       # the common shape stays on the fast path
    assert np.array_equal((rcf[decided] == 1), (rc[decided] == 1))
    ok = decided & (rc == 1)
    assert np.array_equal(out[ok].view(np.uint64), outf[ok].view(np.uint64))
    # and against the standard library on the plausible rows
    for i in np.flatnonzero(ok[:m])[:3000]:
        y, mo, d, h, mi, s, ns, sg, oh, om = (int(v) for v in parts[i])
        t = dt.datetime(y, mo, d, h, mi, s, tzinfo=dt.timezone(dt.timedelta(seconds=sg * (oh * 3600 + om * 60))))
        secs = (t - dt.datetime(1970, 1, 1, tzinfo=dt.timezone.utc)) // dt.timedelta(seconds=1)
        assert out[i] == float(secs * 10 ** 9 + ns) / 1e9
