"""CPU: the host build of the kernels' number parsers (fg_numparse.hpp) against the oracle:
Rust f64::from_str (correct rounding, all three dec2flt stages), serde_json 0.8 numbers, ints."""
import ctypes as C
import struct
import subprocess
from decimal import Decimal, getcontext
from fractions import Fraction
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "native" / "numparse_host.cpp"
LIB = ROOT / "tests" / "native" / "libnumparse_host.so"


@pytest.fixture(scope="module")
def np_lib():
    hdrs = [ROOT / "flowgger_amd/csrc/fg_numparse.hpp", ROOT / "flowgger_amd/csrc/fg_numparse_tables.inc", SRC]
    if not LIB.exists() or LIB.stat().st_mtime < max(h.stat().st_mtime for h in hdrs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math", "-ffp-contract=off",
                        "-o", str(LIB), str(SRC)], check=True)
    L = C.CDLL(str(LIB))
    L.fgn_parse_f64.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.POINTER(C.c_double)]
    L.fgn_json_number.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.fgn_parse_u64.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]
    L.fgn_parse_i64.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_int64)]
    return L


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def ours_f64(L, s, slow=1):
    out = C.c_double()
    b = s.encode()
    rc = L.fgn_parse_f64(b, len(b), slow, C.byref(out))
    return rc, out.value


def check_f64(L, oracle, s, stats=None):
    rc, v = ours_f64(L, s)
    ref = oracle.parse_f64(s)
    if ref is None:
        assert rc == 0, s
    else:
        assert rc == 1 and bits(v) == bits(ref), (s, v, ref)
    if stats is not None and ref is not None:
        rc2, _ = ours_f64(L, s, slow=0)
        stats[rc2] = stats.get(rc2, 0) + 1


def test_f64_grammar_and_specials(np_lib, oracle):
    for s in ["", "+", "-", ".", "e5", "1e", "1e+", " 1", "1 ", "0x10", "1_0", "infin", "nan(1)", "1.e", "--1", "1..2",
              "1e5.5", "inf", "-inf", "+Infinity", "NaN", "-nan", "0", "-0", "0.0", "-0.0", "1", "1.", ".5", "+.5e-3",
              "1e400", "-1e400", "1e-400", "4.9e-324", "2.4703282292062327e-324", "2.4703282292062328e-324",
              "1.7976931348623157e308", "1.7976931348623158e308", "1.7976931348623159e308", "1e23", "8.5e22",
              "9007199254740993", "9007199254740992.5", "9007199254740993e22", "123456789012345678901234567890",
              "0." + "0" * 400 + "1", "1" + "0" * 400, "0" * 100 + "1.5", "1e0000000000000000000000001",
              "1e99999999999999999999", "1e-99999999999999999999", "0e99999999999999999999"]:
        check_f64(np_lib, oracle, s)


def test_f64_random_shapes(np_lib, oracle):
    rng = np.random.default_rng(2022)
    stats = {}
    for i in range(60000):
        nd = int(rng.integers(1, 40)) if i % 5 else int(rng.integers(1, 900))
        digs = "".join(str(int(x)) for x in rng.integers(0, 10, nd))
        k = int(rng.integers(0, 4))
        if k == 0:
            s = digs
        elif k == 1:
            p = int(rng.integers(0, nd + 1))
            s = digs[:p] + "." + digs[p:]
        else:
            p = int(rng.integers(0, nd + 1))
            s = digs[:p] + "." + digs[p:] if k == 3 else digs
            s += "eE"[i % 2] + ["", "+", "-"][int(rng.integers(0, 3))] + str(int(rng.integers(0, 400 if i % 7 else 40)))
        if i % 11 == 0:
            s = "-" + s
        check_f64(np_lib, oracle, s, stats)
    assert stats.get(1, 0) > 10000  # most inputs are decided by stages 1/2


def test_f64_halfway_cases_hit_the_slow_path(np_lib, oracle):
    """Exact midpoints between adjacent doubles, +- one unit far out in the digit string: these
    defeat Eisel-Lemire on truncated mantissas and force the Decimal path."""
    getcontext().prec = 1200
    rng = np.random.default_rng(7)
    stats = {}
    for i in range(1500):
        e = int(rng.integers(1, 2046)) if i % 10 else 0  # include subnormals
        m = int(rng.integers(0, 1 << 52))
        x = struct.unpack("<d", struct.pack("<Q", (e << 52) | m))[0]
        y = struct.unpack("<d", struct.pack("<Q", ((e << 52) | m) + 1))[0]
        if y == float("inf"):
            continue
        mid = (Fraction(x) + Fraction(y)) / 2
        d = Decimal(mid.numerator) / Decimal(mid.denominator)  # exact: denominator is a power of two
        s = format(d, "f") if -300 < d.adjusted() < 300 and i % 2 else format(d, "e")
        for variant in (s, s.replace("e", "1e") if "e" in s else s + "1", ):
            check_f64(np_lib, oracle, variant, stats)
        # just below the midpoint: decrement the last digit of the mantissa digits
        mant = s.split("e")[0] if "e" in s else s
        if mant[-1] in "123456789":
            below = mant[:-1] + str(int(mant[-1]) - 1) + "9" * 30 + (("e" + s.split("e")[1]) if "e" in s else "")
            check_f64(np_lib, oracle, below, stats)
    assert stats.get(2, 0) > 100, stats  # the Decimal path really ran


def test_json_numbers_match_oracle(np_lib, oracle):
    rng = np.random.default_rng(8)
    cases = ["0", "-0", "1", "-1", "01", "1.", ".5", "1e", "1e+", "+1", "-", "1.5e3", "1E-2", "0.0", "-0.0", "0e0",
             "18446744073709551615", "18446744073709551616", "-9223372036854775808", "-9223372036854775809",
             "123456789012345678901234567890", "1.23456789012345678901234567890", "1e308", "1e309", "1e-400",
             "0e99999999999999999999", "1e99999999999999999999", "1e-99999999999999999999", "1385053862.3072",
             "184467440737095516150", "1844674407370955161.5", "0.18446744073709551615", "1e400", "12345678901234567890e-30"]
    for i in range(20000):
        nd = int(rng.integers(1, 30))
        digs = str(int(rng.integers(1, 10))) + "".join(str(int(x)) for x in rng.integers(0, 10, nd - 1))
        s = ("-" if i % 3 == 0 else "") + digs
        if i % 2:
            p = int(rng.integers(1, nd + 1))
            s = s[:len(s) - nd + p] + "." + "".join(str(int(x)) for x in rng.integers(0, 10, int(rng.integers(1, 25))))
        if i % 5 == 0:
            s += "e" + ["", "+", "-"][int(rng.integers(0, 3))] + str(int(rng.integers(0, 330)))
        cases.append(s)
    for s in cases:
        end, kind, b = C.c_uint32(), C.c_uint32(), C.c_uint64()
        ok = np_lib.fgn_json_number(s.encode(), len(s), C.byref(end), C.byref(kind), C.byref(b))
        ref = oracle.json_number(s)
        if ref is None:
            assert not ok or end.value != len(s), s
        else:
            assert ok and end.value == len(s) and (kind.value, b.value) == ref, (s, kind.value, b.value, ref)


def test_integers(np_lib):
    def u(s, mx=2**64 - 1):
        out = C.c_uint64()
        return out.value if np_lib.fgn_parse_u64(s.encode(), len(s), mx, C.byref(out)) else None

    def i(s):
        out = C.c_int64()
        return out.value if np_lib.fgn_parse_i64(s.encode(), len(s), C.byref(out)) else None

    assert u("0") == 0 and u("+7") == 7 and u("007") == 7 and u("18446744073709551615") == 2**64 - 1
    assert u("18446744073709551616") is None and u("") is None and u("+") is None and u("-1") is None and u("1 ") is None
    assert u("255", 255) == 255 and u("256", 255) is None and u("0000256", 255) is None
    assert i("-9223372036854775808") == -2**63 and i("9223372036854775807") == 2**63 - 1
    assert i("9223372036854775808") is None and i("-9223372036854775809") is None and i("-") is None and i("+5") == 5
