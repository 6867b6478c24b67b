"""Build libfg_hip.so (the gfx950 decode kernels + the C ABI of include/fg_hip.h) in-tree.

hipcc cross-compiles for gfx950 without a GPU.  The link step is done by hand so that the
library's DT_NEEDED entry for the HIP runtime is the unversioned ``libamdhip64.so``: inside a
Python process that has imported PyTorch-ROCm, that name resolves to the runtime torch already
loaded (torch ships ``libamdhip64.so`` without a SONAME), so device pointers and streams created
by torch are valid in our kernels; in a standalone host process it resolves through ldconfig to
/opt/rocm.  Two HIP runtimes in one process would not share allocations.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
# FG_BUILD_PROF=1: the MEASUREMENT build (-DFG_PROF_BUILD: s_memtime phase clocks, FG_PROF / FG_ABLATE / FG_PLAN in the
# environment -- csrc/fg_pipeline.hpp) as libfg_hip_prof.so from its own object directory.  tools/ load it with
# FLOWGGER_AMD_PROF_LIB=1; the product library has none of it.
PROF = bool(os.environ.get("FG_BUILD_PROF"))
# FG_BUILD_VARIANT=<name> FG_BUILD_DEFS="-DX -DY": a same-box A/B of COMPILE-TIME choices -- libfg_hip_<name>.so from build_<name>/ with the
# extra defines; tools/ load it with FLOWGGER_AMD_LIB=libfg_hip_<name>.so (flowgger_amd/_lib.py).  Never the product library.
VARIANT = os.environ.get("FG_BUILD_VARIANT", "")
VARIANT_DEFS = os.environ.get("FG_BUILD_DEFS", "").split()
LIB = ROOT / (f"libfg_hip_{VARIANT}.so" if VARIANT else "libfg_hip_prof.so" if PROF else "libfg_hip.so")
ARCH = "gfx950"

HIP_SOURCES = ["fg_rfc5424.hip", "fg_ltsv.hip", "fg_gelf.hip", "fg_frame.hip", "fg_encode.hip", "fg_rfc3164.hip", "fg_calib.hip", "fg_merge.hip"]
HIP_HOST_SOURCES = ["fg_capi.cpp", "fg_host_pipeline.cpp"]  # host code that needs the HIP headers / launch syntax
CXX_SOURCES = ["fg_materialize.cpp", "fg_gather.cpp"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: libfg_hip cannot be built (there is no CPU fallback)")


def _hip_runtime_dir() -> str:
    """Directory holding a libamdhip64.so WITHOUT a versioned SONAME (torch's), else /opt/rocm."""
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            d = Path(spec.origin).parent / "lib"
            if (d / "libamdhip64.so").exists():
                return str(d)
    except Exception:
        pass
    return "/opt/rocm/lib"


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"command failed: {' '.join(cmd[:3])} ...")


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.exists() and d.stat().st_mtime > t for d in deps)


def _depfile_deps(depfile: Path) -> list[Path] | None:
    """Prerequisites recorded by the compiler (-MD -MF) at the last compile of an object; None = unknown."""
    try:
        text = depfile.read_text()
    except OSError:
        return None
    text = text.replace("\\\n", " ")
    _, _, rhs = text.partition(":")
    return [Path(tok) for tok in rhs.split() if tok.startswith(str(ROOT.parent))]


# the object(s) whose kernel a bench workload times.  The launch geometry of the streaming decoders lives in their own sources
# (fg_pipeline.hpp plan_launch); the RFC3164 decoder's and the encoders' tile is picked by fg_tile_cap.hpp.  The host side of the C ABI
# (fg_capi.cpp: contexts, argument checks; fg_host_pipeline.cpp: copies, streams, events) is NOT part of a kernel's identity any
# more (host-side edits disowned every measured HBM-traffic figure).
WORKLOAD_UNITS = {
    "cfg2": ["fg_rfc5424.hip"], "cfg4": ["fg_rfc5424.hip"], "cfg5": ["fg_rfc5424.hip"],
    "cfg1": ["fg_rfc5424.hip", "fg_encode.hip", "fg_tile_cap.hpp"], "frame": ["fg_rfc5424.hip", "fg_frame.hip"],
    "cfg3": ["fg_gelf.hip"], "ltsv": ["fg_ltsv.hip"], "ltsv5": ["fg_ltsv.hip"],
    "rfc3164": ["fg_rfc3164.hip", "fg_tile_cap.hpp"],
}


DEPS_MANIFEST = ROOT / "kernel_deps.json"  # beside the library: travels with it where the build directory does not (GPU box)


def _write_deps_manifest() -> None:
    import json

    units = sorted({u for us in WORKLOAD_UNITS.values() for u in us if not u.endswith(".hpp")})
    man = {}
    for u in units:
        d = _repo_deps(u, manifest=False)
        if d is None:
            return
        man[u] = [str(f.relative_to(ROOT.parent)) for f in d]
    DEPS_MANIFEST.write_text(json.dumps(man, indent=1) + "\n")


def _repo_deps(unit: str, manifest: bool = True) -> list[Path] | None:
    """The repository files the object of `unit` was compiled from (its compiler-written dependency file -- or, where the build
    directory is absent, the manifest build() leaves beside the library), as local paths; None = unknown."""
    try:
        text = (ROOT / "build" / (unit + ".d")).read_text()
    except OSError:
        if not manifest:
            return None
        try:
            import json

            fs = [ROOT.parent / f for f in json.loads(DEPS_MANIFEST.read_text())[unit]]
            return fs if all(f.exists() for f in fs) else None
        except (OSError, KeyError, ValueError):
            return None
    _, _, rhs = text.replace("\\\n", " ").partition(":")
    out = set()
    for tok in rhs.split():
        # (by position in the tree, not by absolute prefix: the snapshot on a GPU box lives under another root)
        for marker, base in (("/flowgger_amd/csrc/", CSRC), ("/include/", ROOT.parent / "include"), ("/tests/native/", ROOT.parent / "tests" / "native")):
            if marker in tok and "/rocm" not in tok and "/usr/" not in tok:
                f = (base / tok.rsplit(marker, 1)[1]).resolve()
                if f.exists():
                    out.add(f)
                break
    return sorted(out) or None


def source_hash(workload: str | None = None) -> str:
    """sha256 (16 hex digits) over the sources a measured figure (profiles/traffic.json) belongs to: with a workload, the files its
    kernel's object was compiled from (so that an edit of the RFC3164 parser does not disown the RFC5424 kernel's PMC figure);
    without one -- or when the dependency files are not there -- every kernel / C-ABI source."""
    import hashlib

    files = None
    if workload in WORKLOAD_UNITS:
        files = []
        for unit in WORKLOAD_UNITS[workload]:
            if unit.endswith(".hpp"):  # a header that is part of the identity by itself
                files += [CSRC / unit]
                continue
            d = _repo_deps(unit)
            if d is None:
                files = None
                break
            # (include/fg_hip.h is the ABI's header, not a kernel's: every added entry point or flag would disown every measured
            #  figure -- round 4 lost them to an enum of the calibration call.  A constant a kernel uses that changes there changes
            #  the kernel's results: the parity tests' business, not this hash's)
            files += [f for f in d if f.name != "fg_hip.h"]
        if files is not None:
            files = sorted(set(files))
    if files is None:
        files = [f for f in sorted(CSRC.glob("*")) + [ROOT.parent / "include" / "fg_hip.h"] if f.suffix in (".hip", ".hpp", ".cpp", ".inc", ".h")]
    h = hashlib.sha256()
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def source_hashes() -> dict:
    return {w: source_hash(w) for w in WORKLOAD_UNITS} | {"*": source_hash()}


def build(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    objdir = ROOT / (f"build_{VARIANT}" if VARIANT else "build_prof" if PROF else "build")
    objdir.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.hpp")) + [ROOT.parent / "include" / "fg_hip.h", Path(__file__)]
    common = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-fast-math",
              "-ffp-contract=off", f"-I{ROOT.parent / 'include'}"]
    # (fg_wave.hpp's host branch includes the fiber emulation from tests/native; the device branch never does, but the
    #  host pass of hipcc still has to find the header)
    common.append(f"-I{ROOT.parent / 'tests' / 'native'}")
    if PROF:
        common.append("-DFG_PROF_BUILD")
    common.extend(VARIANT_DEFS)
    # (source, object, extra defines); fg_encode.hip is compiled once per (encoder, pass) -- its emitters are large
    # force-inlined templates (one object took 18 minutes) -- plus once for the dispatcher
    units: list[tuple[str, Path, list[str]]] = []
    for name in HIP_SOURCES + HIP_HOST_SOURCES + CXX_SOURCES:
        if not (CSRC / name).exists():
            continue
        units.append((name, objdir / (name + ".o"), []))
        if name == "fg_encode.hip":
            for enc in range(5):  # fg_encoder values; GELF (0) has two ranking-scratch sizes
                for wr in (0, 1):
                    for slots in ((1, 8, 32) if enc == 0 else (0,)):
                        units.append((name, objdir / f"fg_encode.e{enc}w{wr}s{slots}.o",
                                      [f"-DFG_ENC_TU={enc}", f"-DFG_ENC_TU_WRITE={wr}", f"-DFG_ENC_TU_SLOTS={slots}"]))
    objs = [u[1] for u in units]

    def compile_unit(unit: tuple[str, Path, list[str]]) -> None:
        name, obj, defs = unit
        src = CSRC / name
        depfile = obj.with_suffix(".d")
        # the headers the object really includes (compiler-written depfile); every header when that is unknown
        deps = _depfile_deps(depfile)
        if not force and not _stale(obj, [src, Path(__file__)] + (deps if deps is not None else headers)):
            return
        md = ["-MD", "-MF", str(depfile)]
        if name in CXX_SOURCES:
            cmd = ["g++", *common, *defs, *md, "-c", str(src), "-o", str(obj)]
        else:
            cmd = [hipcc, f"--offload-arch={ARCH}", "-x", "hip", *common, *defs, *md, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        _run(cmd)

    from concurrent.futures import ThreadPoolExecutor

    jobs = int(os.environ.get("FG_BUILD_JOBS", "0")) or max(1, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        # heaviest first so the long poles start immediately
        order = sorted(units, key=lambda u: 0 if "fg_encode.e0w1" in u[1].name else 1 if "fg_encode.e" in u[1].name else 2)
        for _ in pool.map(compile_unit, order):
            pass
    if os.environ.get("FG_BUILD_NO_LINK"):  # compile only (e.g. while a gpurun snapshot of the tree is in flight)
        return LIB
    if force or _stale(LIB, objs):
        rt = _hip_runtime_dir()
        cmd = ["g++", "-shared", "-o", str(LIB), *map(str, objs), f"-L{rt}", "-lamdhip64",
               "-Wl,--no-undefined", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        _run(cmd)
    if not PROF and not VARIANT:
        _write_deps_manifest()
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
