"""The Rust binding shown in INTEGRATION.md against include/nbody_hip.h, mechanically (no Rust toolchain exists in this
image, so the `extern "C"` text cannot be compiled here: this test is what keeps the two from drifting).

Every `pub fn nbody_*` of every ```rust block must exist in the header with the same arity, the same return type and,
argument by argument, the same shape (pointer depth, constness of the pointee, base type by C ABI class); the two
`#[repr(C)]` structs must list the header's fields in the header's order with the header's types; the constants must equal
the header's enumerators."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RUST_BASE = {"c_int": "i32", "i32": "i32", "i64": "i64", "u32": "u32", "u64": "u64", "u8": "u8", "f32": "f32", "f64": "f64",
             "usize": "usize", "c_char": "char", "c_void": "void",
             "NbodyCtx": "ctx", "NbodyDeltaDecoder": "decoder", "NbodyCounting": "counting", "NbodyParams": "params"}
C_BASE = {"int": "i32", "int32_t": "i32", "int64_t": "i64", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "float": "f32",
          "double": "f64", "size_t": "usize", "char": "char", "void": "void",
          "nbody_ctx": "ctx", "nbody_delta_decoder": "decoder", "nbody_counting": "counting", "nbody_params": "params",
          "nbody_timer": "timer", "nbody_host_tree": "host_tree", "nbody_tree_view": "tree_view"}


def _strip_c_comments(txt):
    return re.sub(r"/\*.*?\*/", "", txt, flags=re.S)


def c_type(t):
    """'const float*' -> ('f32', depth 1, pointee const)."""
    t = t.strip()
    depth = t.count("*")
    const = bool(re.search(r"\bconst\b", t))
    base = re.sub(r"\bconst\b|\bstruct\b|\*", " ", t).split()
    assert len(base) == 1, t
    return C_BASE[base[0]], depth, const


def rust_type(t):
    t = t.strip()
    depth, const = 0, False
    first = True
    while True:
        m = re.match(r"\*(mut|const)\s+(.*)", t)
        if not m:
            break
        if first:
            const = m.group(1) == "const"     # constness of what the OUTER pointer points at matters for `T*` vs `const T*`
        # for pointer-to-pointer (nbody_ctx**) the header has no const at all
        first = False
        depth += 1
        t = m.group(2).strip()
    return RUST_BASE[t], depth, const


def header_functions():
    txt = _strip_c_comments(open(os.path.join(ROOT, "include", "nbody_hip.h")).read())
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(nbody_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        arglist = []
        if args and args != "void":
            for a in args.split(","):
                a = re.sub(r"^(.*?)(\b\w+)\s*\[\d*\]$", r"\1* \2", a.strip())     # `int32_t out[4]` is `int32_t* out`
                mm = re.match(r"(.*?)(\b[a-zA-Z_]\w*)$", a)     # drop the parameter name
                arglist.append(c_type(mm.group(1)))
        out[name] = (None if ret == "void" else c_type(ret), arglist)
    return out, txt


def rust_blocks():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return "\n".join(re.findall(r"```rust\n(.*?)```", md, flags=re.S))


def rust_functions():
    out = {}
    for m in re.finditer(r"pub fn (nbody_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", rust_blocks()):
        name, args, ret = m.group(1), m.group(2).strip(), m.group(3)
        arglist = [rust_type(a.split(":", 1)[1]) for a in args.split(",") if a.strip()] if args else []
        out[name] = (None if ret is None else rust_type(ret), arglist)
    return out


def test_the_binding_declares_real_functions_with_the_headers_signatures():
    hdr, _ = header_functions()
    rust = rust_functions()
    assert len(rust) >= 20, "the extern block of INTEGRATION.md was not found"
    for name, (rret, rargs) in rust.items():
        assert name in hdr, f"INTEGRATION.md binds {name}, which include/nbody_hip.h does not declare"
        cret, cargs = hdr[name]
        assert len(rargs) == len(cargs), f"{name}: {len(rargs)} arguments in the binding, {len(cargs)} in the header"
        if cret is None or rret is None:
            assert cret is None and rret is None, f"{name}: return type"
        else:
            assert rret[:2] == cret[:2] and (cret[1] == 0 or rret[2] == cret[2]), f"{name}: returns {rret} vs header {cret}"
        for k, (ra, ca) in enumerate(zip(rargs, cargs)):
            assert ra[0] == ca[0] or {ra[0], ca[0]} == {"u8", "void"}, f"{name} argument {k}: base type {ra[0]} vs header {ca[0]}"
            assert ra[1] == ca[1], f"{name} argument {k}: pointer depth {ra[1]} vs header {ca[1]}"
            if ca[1] == 1:
                assert ra[2] == ca[2], f"{name} argument {k}: {'*const' if ra[2] else '*mut'} vs header {'const' if ca[2] else 'mutable'}"


def _c_struct_fields(txt, name):
    m = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", txt, flags=re.S)
    assert m, name
    return [(f.split()[-1], C_BASE[f.split()[0]]) for f in (x.strip() for x in m.group(1).split(";")) if f]


def _rust_struct_fields(name):
    m = re.search(r"pub struct " + name + r"\s*\{(.*?)\}", rust_blocks(), flags=re.S)
    assert m, name
    return [(f.split(":")[0].replace("pub", "").strip(), RUST_BASE[f.split(":")[1].strip()]) for f in m.group(1).split(",") if ":" in f]


@pytest.mark.parametrize("rust_name,c_name", [("NbodyCounting", "nbody_counting"), ("NbodyParams", "nbody_params")])
def test_repr_c_structs_match_field_for_field(rust_name, c_name):
    _, txt = header_functions()
    assert _rust_struct_fields(rust_name) == _c_struct_fields(txt, c_name)
    src = rust_blocks()
    assert re.search(r"#\[repr\(C\)\][^\n]*\n?\s*pub struct " + rust_name, src), f"{rust_name} must be #[repr(C)]"


def test_constants_equal_the_headers_enumerators():
    _, txt = header_functions()
    enums = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(NBODY_[A-Z_0-9]+)\s*=\s*(-?\d+)", txt)}
    enums.update({m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(NBODY_[A-Z_0-9]+)\s+(-?\d+)", txt)})
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (NBODY_[A-Z_0-9]+): c_int = (-?\d+);", rust_blocks())}
    assert consts, "no constants found in the binding"
    for k, v in consts.items():
        assert enums.get(k) == v, f"{k} = {v} in the binding, {enums.get(k)} in the header"
