#!/usr/bin/env python3
"""Generate the Rust FFI crate source (integration/rust/fg-hip-sys/src/lib.rs) from include/fg_hip.h -- what bindgen would
do, without needing libclang or a Rust toolchain (neither is in the build image).  `--check` regenerates in memory and fails
when the committed file differs (tests/test_rust_ffi_cpu.py runs it), so the crate cannot drift from the header."""
from __future__ import annotations

import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "fg_hip.h"
OUT = ROOT / "integration" / "rust" / "fg-hip-sys" / "src" / "lib.rs"

BASE = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int8_t": "i8", "int32_t": "i32", "int64_t": "i64",
        "int": "c_int", "double": "f64", "float": "f32", "char": "c_char", "void": "c_void"}


def strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def rust_type(ctype: str, known: set[str]) -> str:
    """C declarator type (no identifier) -> Rust."""
    toks = re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\*", ctype)
    # walk left to right: [const] base [const] then a chain of '*' each optionally followed by const
    i, const_base = 0, False
    if toks[i] == "const":
        const_base, i = True, i + 1
    if toks[i] in ("struct", "enum"):
        i += 1
    base = toks[i]
    i += 1
    if i < len(toks) and toks[i] == "const":
        const_base, i = True, i + 1
    rt = BASE.get(base, base if base in known else None)
    if rt is None:
        raise ValueError(f"unknown C type {base!r} in {ctype!r}")
    pointee_const = const_base
    while i < len(toks):
        assert toks[i] == "*", ctype
        i += 1
        rt = f"*const {rt}" if pointee_const else f"*mut {rt}"
        pointee_const = False
        if i < len(toks) and toks[i] == "const":
            pointee_const, i = True, i + 1
    return rt


def split_decl(decl: str) -> tuple[str, str]:
    """'const uint8_t* d_bytes' -> ('const uint8_t*', 'd_bytes'); 'uint64_t sizes[FG_TABLE_ARRAYS]' -> ('uint64_t*', 'sizes')"""
    decl = decl.strip()
    arr = re.search(r"\[[^\]]*\]\s*$", decl)
    if arr:
        decl = decl[:arr.start()].strip()
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z_0-9]*)$", decl, flags=re.S)
    ctype, name = m.group(1).strip(), m.group(2)
    if arr:
        ctype += "*"
    return ctype, name


RUST_KEYWORDS = {"final": "is_final", "type": "type_", "in": "in_", "ref": "ref_", "match": "match_"}


def parse(text: str):
    """-> (defines, enums, structs, opaque, functions) in header order."""
    src = strip_comments(text)
    body = src[src.index('extern "C" {') + len('extern "C" {'):]
    items = []
    pos = 0
    # object-like and function-like macros (whole header, they precede / interleave with the declarations)
    for m in re.finditer(r"^[ \t]*#define[ \t]+([A-Za-z_][A-Za-z_0-9]*)(\([^)]*\))?[ \t]+(.+?)[ \t]*$", src, flags=re.M):
        items.append((m.start(), "define", (m.group(1), m.group(2), m.group(3))))
    off = src.index('extern "C" {') + len('extern "C" {')
    for m in re.finditer(r"typedef\s+enum\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", body, flags=re.S):
        items.append((off + m.start(), "enum", (m.group(3), m.group(2))))
    for m in re.finditer(r"(?<!typedef\s)(?<![A-Za-z_0-9])enum\s*\{(.*?)\}\s*;", body, flags=re.S):
        items.append((off + m.start(), "enum", (None, m.group(1))))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", body, flags=re.S):
        items.append((off + m.start(), "struct", (m.group(3), m.group(2))))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", body):
        items.append((off + m.start(), "opaque", m.group(2)))
    # prototypes: at file scope of the extern block, "ret name(args);"
    flat = re.sub(r"typedef\s+(enum|struct)\s+\w+\s*\{.*?\}\s*\w+\s*;", lambda m: " " * len(m.group(0)), body, flags=re.S)
    flat = re.sub(r"enum\s*\{.*?\}\s*;", lambda m: " " * len(m.group(0)), flat, flags=re.S)
    flat = re.sub(r"^[ \t]*#.*$", lambda m: " " * len(m.group(0)), flat, flags=re.M)
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9\s\*]*?)\b(\w+)\s*\(([^;{}]*)\)\s*;", flat, flags=re.S):
        items.append((off + m.start(), "fn", (m.group(1).strip(), m.group(2), m.group(3).strip())))
    items.sort(key=lambda t: t[0])
    return items


def eval_c_int(expr: str, env: dict[str, int]) -> int:
    e = re.sub(r"(?<=[0-9a-fA-Fx])[uUlL]+\b", "", expr.strip())
    e = re.sub(r"\b([A-Za-z_]\w*)\b", lambda m: str(env[m.group(1)]) if m.group(1) in env else m.group(1), e)
    return int(eval(e, {"__builtins__": {}}))  # integer literals and + - << | only (our own header)


def generate() -> str:
    items = parse(HEADER.read_text())
    known = {it[2][0] for it in items if it[1] in ("struct", "enum") and it[2][0]} | {it[2] for it in items if it[1] == "opaque"}
    out = []
    w = out.append
    w("// fg-hip-sys: raw FFI declarations of libfg_hip.so (include/fg_hip.h, FG_ABI_VERSION 3).")
    w("// GENERATED by tools/gen_rust_ffi.py from the header -- do not edit; `python tools/gen_rust_ffi.py` rewrites it and")
    w("// tests/test_rust_ffi_cpu.py fails when this file and the header disagree.")
    w("//")
    w("// Reference interfaces behind these entry points (flowgger source tree): trait Decoder, src/flowgger/decoder/mod.rs:23-46;")
    w("// Record, src/flowgger/record.rs:70-82; the per-line call site the batching framer replaces,")
    w("// src/flowgger/splitter/line_splitter.rs:44-54.  The safe wrapper lives in ../flowgger/decoder/gpu_decoder.rs.")
    w("#![allow(non_camel_case_types, non_upper_case_globals, non_snake_case, dead_code)]")
    w("use std::os::raw::{c_char, c_int, c_void};")
    w("")
    env: dict[str, int] = {}
    fns = []
    for _, kind, val in items:
        if kind == "define":
            name, params, body = val
            if name in ("FG_HIP_H",):
                continue
            if params:  # FG_META_*(m): accessor macros -> const fns
                m = re.match(r"\(\(uint8_t\)\(\(?\(m\)\s*(?:>>\s*(\d+)\)?)?\s*&\s*0xFF\)\)", body)
                sh = int(m.group(1)) if m and m.group(1) else 0
                w(f"#[inline] pub const fn {name}(m: u32) -> u8 {{ ((m >> {sh}) & 0xFF) as u8 }}")
            elif name == "FG_STREAM_OWN":
                w("/// `stream` argument: the ctx's own non-blocking stream")
                w("pub const FG_STREAM_OWN: *mut c_void = usize::MAX as *mut c_void;")
            else:
                v = eval_c_int(body, env)
                env[name] = v
                ty = "u32" if body.strip().lower().endswith("u") or v > 0x7FFFFFFF else "c_int"
                if name in ("FG_ST_OVERFLOW", "FG_ST_BAD_UTF8"):
                    ty = "u8"
                w(f"pub const {name}: {ty} = 0x{v:X};" if v > 255 else f"pub const {name}: {ty} = {v};")
        elif kind == "enum":
            tname, body = val
            if tname:
                w(f"pub type {tname} = c_int;")
            nxt = 0
            for ent in [e.strip() for e in body.split(",") if e.strip()]:
                if "=" in ent:
                    k, v = [x.strip() for x in ent.split("=", 1)]
                    nxt = eval_c_int(v, env)
                else:
                    k = ent
                env[k] = nxt
                w(f"pub const {k}: {tname or 'c_int'} = {nxt};")
                nxt += 1
            w("")
        elif kind == "opaque":
            w(f"#[repr(C)] pub struct {val} {{ _private: [u8; 0] }}  // opaque: only ever behind a pointer")
            w("")
        elif kind == "struct":
            name, body = val
            w("#[repr(C)]")
            w("#[derive(Clone, Copy)]")
            w(f"pub struct {name} {{")
            for field in [f.strip() for f in body.split(";") if f.strip()]:
                # "uint32_t off" | "const char* const* keys" | several declarators are not used in this header
                ctype, fname = split_decl(field)
                w(f"    pub {RUST_KEYWORDS.get(fname, fname)}: {rust_type(ctype, known)},")
            w("}")
            w("")
        elif kind == "fn":
            fns.append(val)
    w('#[link(name = "fg_hip")]')
    w('extern "C" {')
    for ret, name, args in fns:
        params = []
        if args and args != "void":
            for a in [x.strip() for x in args.split(",")]:
                ctype, pname = split_decl(a)
                params.append(f"{RUST_KEYWORDS.get(pname, pname)}: {rust_type(ctype, known)}")
        r = "" if ret == "void" else f" -> {rust_type(ret, known)}"
        line = f"    pub fn {name}({', '.join(params)}){r};"
        if len(line) > 130:
            line = f"    pub fn {name}(\n        " + ",\n        ".join(params) + f",\n    ){r};"
        w(line)
    w("}")
    w("")
    return "\n".join(out)


def main() -> int:
    text = generate()
    if "--check" in sys.argv:
        if not OUT.exists() or OUT.read_text() != text:
            print(f"{OUT} is out of date with {HEADER}: run python tools/gen_rust_ffi.py", file=sys.stderr)
            return 1
        return 0
    OUT.parent.mkdir(parents=True, exist_ok=True)
    OUT.write_text(text)
    print(f"wrote {OUT} ({len(text.splitlines())} lines)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
