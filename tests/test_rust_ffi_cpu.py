"""CPU: the Rust side of the boundary (integration/rust/) against include/fg_hip.h.

rustc is not in the build image, so the crate cannot be compiled here; what CAN be machine-checked is that the FFI
declarations a maintainer would compile agree with the C header and with the library:
  * fg-hip-sys/src/lib.rs is exactly what tools/gen_rust_ffi.py derives from include/fg_hip.h (no drift);
  * every #[repr(C)] struct has the size and field offsets gcc gives the C struct (Rust repr(C) layout rules applied to
    the declared field types), every constant has the value the C compiler sees;
  * every extern "C" fn is exported by libfg_hip.so, with as many parameters as the header's prototype;
  * the hand-written safe layer (gpu_decoder.rs, batching_splitter.rs) only uses items the crate declares, and implements
    the reference's trait (decoder/mod.rs:44-46) and Record fields (record.rs:70-82) by name."""
import ctypes as C
import re
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
LIB_RS = ROOT / "integration" / "rust" / "fg-hip-sys" / "src" / "lib.rs"
HEADER = ROOT / "include" / "fg_hip.h"
SAFE = [ROOT / "integration" / "rust" / "flowgger" / "decoder" / "gpu_decoder.rs",
        ROOT / "integration" / "rust" / "flowgger" / "splitter" / "batching_splitter.rs"]

SIZES = {"u8": (1, 1), "i8": (1, 1), "u32": (4, 4), "i32": (4, 4), "c_int": (4, 4), "f32": (4, 4), "u64": (8, 8), "i64": (8, 8),
         "f64": (8, 8)}


def rust_items():
    src = LIB_RS.read_text()
    consts = {m.group(1): m.group(3) for m in re.finditer(r"pub const (\w+): ([\w\* ]+) = ([^;]+);", src)}
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\}", src, flags=re.S):
        fields = re.findall(r"pub (\w+): ([^,]+),", m.group(2))
        if fields:
            structs[m.group(1)] = fields
    ext = src[src.index('extern "C" {'):]
    fns = {}
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", ext, flags=re.S):
        params = [p.strip() for p in m.group(2).split(",") if p.strip()]
        fns[m.group(1)] = (params, (m.group(3) or "").strip())
    return src, consts, structs, fns


def test_crate_is_what_the_header_generates():
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def layout(fields, structs):
    """repr(C): each field at the next multiple of its alignment; size rounded up to the struct's alignment."""
    off, align, offs = 0, 1, []
    for _, ty in fields:
        ty = ty.strip()
        if ty.startswith("*"):
            s, a = 8, 8
        elif ty in SIZES:
            s, a = SIZES[ty]
        elif ty in structs:
            s, a, _ = layout(structs[ty], structs)
        elif ty in ("fg_format", "fg_framing", "fg_encoder", "fg_merger"):
            s, a = 4, 4
        else:
            raise AssertionError(f"unknown Rust field type {ty}")
        off = (off + a - 1) // a * a
        offs.append(off)
        off += s
        align = max(align, a)
    return (off + align - 1) // align * align, align, offs


def test_struct_layouts_and_constants_match_the_c_compiler(tmp_path):
    _, consts, structs, _ = rust_items()
    c_field = {"final": "final", "is_final": "final", "type_": "type"}
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for name, fields in structs.items():
        prog.append(f'  printf("S {name} %zu\\n", sizeof({name}));')
        for f, _ in fields:
            prog.append(f'  printf("F {name}.{f} %zu\\n", offsetof({name}, {c_field.get(f, f)}));')
    for name in consts:
        if name != "FG_STREAM_OWN":
            prog.append(f'  printf("C {name} %lld\\n", (long long){name});')
    prog.append("  return 0; }")
    (tmp_path / "abi.c").write_text("\n".join(prog))
    subprocess.run(["gcc", "-std=c11", "-o", str(tmp_path / "abi"), str(tmp_path / "abi.c")], check=True)
    seen = {}
    for line in subprocess.run([str(tmp_path / "abi")], capture_output=True, text=True, check=True).stdout.splitlines():
        kind, key, val = line.split()
        seen[(kind, key)] = int(val)
    assert len(structs) >= 8 and "fg_tables" in structs and "fg_transcoded" in structs
    for name, fields in structs.items():
        size, _, offs = layout(fields, structs)
        assert seen[("S", name)] == size, f"sizeof({name}): C {seen[('S', name)]} != Rust repr(C) {size}"
        for (f, _), o in zip(fields, offs):
            assert seen[("F", f"{name}.{f}")] == o, f"offsetof({name}, {f})"
    for name, val in consts.items():
        if name == "FG_STREAM_OWN":
            continue
        v = int(val.replace("_", ""), 0)
        c = seen[("C", name)]
        assert v == c or (v & 0xFFFFFFFF) == (c & 0xFFFFFFFF), f"{name}: Rust {v} != C {c}"


def test_every_extern_fn_is_exported_with_the_headers_arity():
    from flowgger_amd import _lib

    lib = C.CDLL(str(_lib.LIB_PATH))
    _, _, _, fns = rust_items()
    hdr = re.sub(r"/\*.*?\*/", " ", HEADER.read_text(), flags=re.S)
    protos = {m.group(1): m.group(2) for m in re.finditer(r"\b(fg_\w+)\s*\(([^;{}]*)\)\s*;", hdr)}
    assert set(fns) == set(protos), set(fns) ^ set(protos)
    assert len(fns) >= 27
    for name, (params, _ret) in fns.items():
        assert hasattr(lib, name), f"libfg_hip.so does not export {name}"
        args = protos[name].strip()
        arity = 0 if args in ("", "void") else len(args.split(","))
        assert len(params) == arity, name


def test_safe_layer_uses_only_declared_items_and_keeps_the_reference_interface():
    src, consts, structs, fns = rust_items()
    declared = set(consts) | set(structs) | set(fns) | set(re.findall(r"pub type (\w+)", src)) | \
        set(re.findall(r"pub const fn (\w+)", src)) | {"fg_ctx", "fg_hip_sys"}
    for f in SAFE:
        text = f.read_text()
        code = "\n".join(ln for ln in text.splitlines() if not ln.lstrip().startswith("//"))
        used = set(re.findall(r"\b(fg_\w+|FG_[A-Z0-9_]+)\b", code))
        assert used - declared == set(), f"{f.name} uses undeclared items: {sorted(used - declared)}"
        # calls pass as many arguments as the prototype has parameters
        for m in re.finditer(r"\b(fg_\w+)\(", code):
            name = m.group(1)
            if name not in fns or code[max(0, m.start() - 7):m.start()].endswith("pub fn "):
                continue
            depth, i, args, cur = 1, m.end(), [], ""
            while depth:
                ch = code[i]
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                    if depth == 0:
                        break
                if ch == "," and depth == 1:
                    args.append(cur)
                    cur = ""
                else:
                    cur += ch
                i += 1
            if cur.strip():
                args.append(cur)
            assert len(args) == len(fns[name][0]), f"{f.name}: {name} called with {len(args)} arguments, declared with {len(fns[name][0])}"
    dec = SAFE[0].read_text()
    assert "impl Decoder for GpuDecoder" in dec and "fn decode(&self, line: &str) -> Result<Record, &'static str>" in dec
    assert "impl Clone for GpuDecoder" in dec and "unsafe impl Send for GpuDecoder" in dec and "impl Drop for GpuDecoder" in dec
    for field in ("ts", "hostname", "facility", "severity", "appname", "procid", "msgid", "msg", "full_msg", "sd"):  # record.rs:70-82
        assert re.search(rf"\b{field}[:,]", dec[dec.index("Ok(Record {"):]), field
