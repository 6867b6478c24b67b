import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


# The CPU suite builds its native helpers (tests/native/lib*_host.so, the fake-runtime host pipeline ...) on first use, each from the
# fixture of the module that needs it.  Under pytest-xdist several workers reach the same fixture at the same moment and compiled INTO
# THE SAME FILE -- a worker then loaded a half-written library ("file too short").  Every g++ / gcc build whose output lies under
# tests/native therefore compiles to a private name and is renamed into place (atomic on one filesystem).
import os
import subprocess

_real_run = subprocess.run


def _atomic_native_build(cmd, *args, **kwargs):
    if isinstance(cmd, (list, tuple)) and cmd and cmd[0] in ("g++", "gcc") and "-o" in cmd:
        i = list(cmd).index("-o")
        out = Path(cmd[i + 1])
        if str(out.resolve()).startswith(str(ROOT / "tests" / "native")):
            tmp = out.with_name(f".{out.name}.{os.getpid()}.tmp")
            c2 = list(cmd)
            c2[i + 1] = str(tmp)
            r = _real_run(c2, *args, **kwargs)
            if r.returncode == 0:
                os.replace(tmp, out)
            return r
    return _real_run(cmd, *args, **kwargs)


subprocess.run = _atomic_native_build


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a machine without an MI355X: the gpu-marked tests are skipped (with the reason), not failed with
    FG_ERR_NO_DEVICE.  On a box WITH a GPU nothing is skipped: a missing libfg_hip.so or a failing kernel fails loudly."""
    try:
        import torch

        have_gpu = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False); run through gpurun with -m gpu")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding

    return oracle_binding.Oracle()


def pytest_sessionstart(session):
    """libfg_hip.so is a build product (git-ignored): when it is missing -- a fresh checkout on which
    __graft_entry__.build() has not run yet -- compile it once (hipcc cross-compiles gfx950 without a GPU,
    under a minute on 8 cores).  Without hipcc the tests that load the library fail loudly, as they should."""
    from flowgger_amd import build as fg_build

    if not fg_build.LIB.exists():
        try:
            fg_build.build()
        except Exception as e:  # noqa: BLE001
            print(f"[conftest] libfg_hip.so is missing and could not be built: {e}", file=sys.stderr)
