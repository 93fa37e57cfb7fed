#!/usr/bin/env python3
"""Generates rust/bzk-sys/src/lib.rs - the raw `extern "C"` view of include/bzk.h for a Rust host - mechanically from the header:
every function prototype, every `typedef struct { .. }` as #[repr(C)] (same field order), every opaque handle, every #define'd
integer.  tests/test_rust_shim_cpu.py re-runs this generator and compares its output with the committed file, so the Rust
declarations cannot drift from the C ABI unnoticed.  (No rustc in this image: the crate is shipped as source, unverified by
compilation - the check above is what replaces the compiler for the FFI surface.)

usage: python tools/gen_rust_sys.py [--check]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bzk.h")
OUT = os.path.join(ROOT, "rust", "bzk-sys", "src", "lib.rs")

SCALARS = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "int64_t": "i64", "double": "f64", "char": "c_char",
           "void": "c_void", "int": "i32"}


def strip_comments(src: str) -> str:
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def rust_type(ctype: str, opaque) -> str:
    """C parameter / field / return type (declarator suffixes like [32] already folded into a pointer) -> Rust"""
    t = " ".join(ctype.replace("*", " * ").split())
    toks = t.split()
    # base type (with a possible leading const) followed by pointer levels, each possibly const-qualified
    const_base = False
    i = 0
    if toks[i] == "const":
        const_base = True
        i += 1
    if toks[i] == "struct":
        i += 1
    base = toks[i]
    i += 1
    if i < len(toks) and toks[i] == "const":  # `T const`
        const_base = True
        i += 1
    rb = SCALARS.get(base) or (base if base in opaque else None)
    if rb is None:
        raise ValueError(f"unknown C type {base!r} in {ctype!r}")
    levels = []  # constness of what each pointer level points AT, innermost first
    cur_const = const_base
    while i < len(toks):
        assert toks[i] == "*", ctype
        levels.append(cur_const)
        i += 1
        cur_const = False
        if i < len(toks) and toks[i] == "const":
            cur_const = True
            i += 1
    out = rb
    for c in levels:
        out = ("*const " if c else "*mut ") + out
    return out


def parse(src: str):
    src = strip_comments(src)
    consts = [(m.group(1), m.group(2)) for m in re.finditer(r"^#define\s+(BZK_[A-Z0-9_]+)\s+\(?(-?\d+)u?\)?\s*$", src, flags=re.M)]
    body = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    body = body.replace('extern "C" {', "").strip()
    opaque, structs, funcs = [], [], []
    # typedef struct { ... } name;
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", body, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # `const uint8_t *deposit_vk` / `uint8_t a, b, c` / `uint32_t n_in, n_aux`
            mm = re.match(r"(.*?)([\w\s,\*]+)$", decl)
            head, names = decl.rsplit(" ", 1) if "," not in decl else (None, None)
            if "," in decl:
                first, rest = decl.split(",", 1)
                ty, n0 = first.rsplit(" ", 1)
                names = [n0] + [x.strip() for x in rest.split(",")]
                for n in names:
                    fields.append((n.lstrip("*"), ty + (" *" if n.startswith("*") else "")))
            else:
                ty, n = decl.rsplit(" ", 1)
                stars = len(n) - len(n.lstrip("*"))
                fields.append((n.lstrip("*"), ty + " *" * stars))
        structs.append((m.group(2), fields))
    body_nostruct = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", body, flags=re.S)
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", body_nostruct):
        opaque.append(m.group(2))
    body_nostruct = re.sub(r"typedef\s+struct\s+\w+\s+\w+\s*;", "", body_nostruct)
    names = set(opaque) | {s[0] for s in structs}
    for stmt in body_nostruct.split(";"):
        stmt = " ".join(stmt.split())
        m = re.match(r"^(.*?)\b(bzk_\w+)\s*\((.*)\)$", stmt)
        if not m:
            continue
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = re.search(r"\[\w*\]\s*$", a)
                if arr:
                    a = a[: arr.start()].strip()
                mm = re.match(r"^(.*?)(\w+)$", a)
                ty, pn = mm.group(1).strip(), mm.group(2)
                if arr:
                    ty += " *"
                params.append((pn, ty))
        funcs.append((name, ret, params))
    return consts, opaque, structs, funcs, names


RUST_KEYWORDS = {"in", "type", "ref", "box", "loop", "match", "mod", "move", "self", "use", "where", "fn", "impl", "as"}


def ident(n: str) -> str:
    return n + "_" if n in RUST_KEYWORDS else n


def generate() -> str:
    consts, opaque, structs, funcs, names = parse(open(HEADER).read())
    o = []
    o.append("// GENERATED by tools/gen_rust_sys.py from include/bzk.h - do not edit; re-run the generator instead.")
    o.append("// Raw FFI surface of libbzk.so (MI355X-native Groth16 hot path for Bazuka's MPN rollup).  UNVERIFIED BY COMPILATION: the")
    o.append("// image that builds libbzk has no rustc; tests/test_rust_shim_cpu.py checks this file against the header mechanically.")
    o.append("#![allow(non_camel_case_types, non_upper_case_globals, dead_code)]")
    o.append("use std::os::raw::{c_char, c_void};")
    o.append("")
    for k, v in consts:
        ty = "i32" if k.startswith("BZK_E_") or k == "BZK_OK" or k.startswith("BZK_REFUSE_") else ("usize" if k.endswith("_BYTES") else "u32")
        o.append(f"pub const {k}: {ty} = {v};")
    o.append("")
    for n in opaque:
        o.append(f"#[repr(C)] pub struct {n} {{ _private: [u8; 0] }}")
    o.append("")
    for n, fields in structs:
        o.append("#[repr(C)]")
        o.append("#[derive(Clone, Copy)]")
        o.append(f"pub struct {n} {{")
        for fn_, ty in fields:
            o.append(f"    pub {ident(fn_)}: {rust_type(ty, names)},")
        o.append("}")
        o.append("")
    o.append('#[link(name = "bzk")]')
    o.append('extern "C" {')
    for name, ret, params in funcs:
        ps = ", ".join(f"{ident(pn)}: {rust_type(ty, names)}" for pn, ty in params)
        r = "" if ret == "void" else f" -> {rust_type(ret, names)}"
        o.append(f"    pub fn {name}({ps}){r};")
    o.append("}")
    o.append("")
    return "\n".join(o)


def main():
    text = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("rust/bzk-sys/src/lib.rs is stale: run python tools/gen_rust_sys.py")
            sys.exit(1)
        print("rust/bzk-sys/src/lib.rs is up to date")
        return
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {text.count('pub fn ')} functions")


if __name__ == "__main__":
    main()
