"""CPU: the framers' flush policy (flowgger_amd/host/fg_decoder.hpp FlushPolicy / BufferedSource), VERDICT r2 "a framer that flushes on
time": a connection that delivers a few lines and then stalls gets them decoded within the latency bound -- not when 8 MiB have
arrived -- and the idle timeout ends the connection the way line_splitter.rs:26-33 does.  The GPU call is a counting stand-in here;
tests/test_gpu_round3.py runs the real splitters over a pipe."""
import subprocess
import time
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("framer") / "framer_policy_test"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", str(ROOT / "tests/native/framer_policy_test.cpp"), "-o", str(out)], check=True)
    return out


def run(exe, mode, idle_ms, latency_ms, script, extra=()):
    p = subprocess.Popen([str(exe), mode, str(idle_ms), str(latency_ms), *map(str, extra)], stdin=subprocess.PIPE, stdout=subprocess.PIPE)
    t0 = time.monotonic()
    for delay, payload in script:
        time.sleep(delay)
        if payload is None:
            p.stdin.close()
        else:
            p.stdin.write(payload)
            p.stdin.flush()
    out = p.stdout.read().decode().split("\n")
    p.wait(timeout=20)
    return [ln.split() for ln in out if ln], time.monotonic() - t0


def test_lines_that_arrived_are_flushed_when_the_source_stalls(exe):
    ten = b"".join(b"line %d\n" % i for i in range(10))
    ev, _ = run(exe, "line", 700, 20, [(0.05, ten), (0.30, b"a\nb\n"), (0.004, b"c\nd\ne\n"), (0.25, b"tail without newline")])
    flushes = [(int(e[1]), int(e[2])) for e in ev if e[0] == "FLUSH"]
    assert [n for n, _ in flushes] == [10, 5], ev  # the burst split over two writes 4 ms apart is ONE batch (linger)
    assert flushes[0][1] < 50 + 20 + 80, ev          # 10 lines out ~max_latency after they arrived, not at 8 MiB / EOF
    assert flushes[1][1] - 350 < 20 + 100, ev
    assert ev[-1] == ["END", "idle"], ev             # no data for idle_timeout: the connection closes; the partial line is dropped


def test_eof_flushes_the_unterminated_tail_and_a_big_batch_does_not_linger(exe):
    ev, _ = run(exe, "line", -1, 200, [(0.02, b"x\n" * 40000), (0.05, b"last"), (0.01, None)], extra=[1024])
    flushes = [(int(e[1]), int(e[2])) for e in ev if e[0] == "FLUSH"]
    assert sum(n for n, _ in flushes) == 40001 and ev[-1] == ["END", "eof"], ev[-3:]
    assert flushes[0][1] < 150, ev[:3]  # 80 KB > linger_below: flushed as soon as the source ran dry, without the 200 ms linger


def test_chunk_mode_hands_over_what_has_arrived(exe):
    ev, _ = run(exe, "chunk", 500, 10, [(0.02, b"a" * 1000), (0.2, b"b" * 3000), (0.002, b"c" * 500)])
    chunks = [(int(e[1]), int(e[2])) for e in ev if e[0] == "CHUNK"]
    assert [n for n, _ in chunks] == [1000, 3500], ev
    assert chunks[0][1] < 20 + 10 + 80 and ev[-1] == ["END", "idle"], ev
