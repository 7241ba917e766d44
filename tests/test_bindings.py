"""The Rust declarations in bindings/rust/rt-amd-sys must say what include/rt_amd.h says:
same functions with the same argument types, same struct fields in the same order with
equivalent types, same constants. (There is no Rust toolchain in the build image, so the crate
itself cannot be compiled here; this keeps it from drifting.)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_amd.h")
LIB_RS = os.path.join(ROOT, "bindings", "rust", "rt-amd-sys", "src", "lib.rs")

C_TO_RUST = {
    "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "uint8_t": "u8", "double": "f64", "float": "f32",
    "int": "c_int", "size_t": "usize", "char": "c_char", "void": "c_void",
}


def strip_c(text):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))


def rust_type_of(ctype, array=None):
    """'const double*' -> '*const f64'; 'rt_scene**' -> '*mut *mut rt_scene'."""
    t = ctype.strip()
    stars = t.count("*")
    t = t.replace("*", " ").split()
    const = "const" in t
    base = [w for w in t if w not in ("const", "struct")][0]
    base = C_TO_RUST.get(base, base)
    out = base
    for i in range(stars):
        # only the innermost pointee can be const in this header
        out = ("*const " if (const and i == 0) else "*mut ") + out
    if array:
        out = f"[{out}; {array}]"
    return out


def header_structs():
    text = strip_c(open(HEADER).read())
    out = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\}\s*(\w+);", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r"(.+?)\s*(\w+)(?:\[(\d+)\])?$", decl)
            ctype, fname, arr = m.group(1), m.group(2), m.group(3)
            fields.append((fname, rust_type_of(ctype, arr)))
        out[name] = fields
    return out


def header_functions():
    text = strip_c(open(HEADER).read())
    text = re.sub(r"typedef struct \w+ \{.*?\}\s*\w+;", "", text, flags=re.S)
    text = re.sub(r"^[ \t]*#.*$", "", text, flags=re.M)  # preprocessor lines
    out = {}
    for ret, name, args in re.findall(r"([\w\s\*]+?)\b(rt_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        params = []
        if args.strip() != "void":
            for a in args.split(","):
                m = re.match(r"(.+?)\s*(\w+)(\[\d*\])?$", a.strip())
                ctype = m.group(1) + ("*" if m.group(3) else "")  # array parameters decay to pointers
                params.append(rust_type_of(ctype))
        ret = ret.strip()
        out[name] = (params, None if ret == "void" else rust_type_of(ret))
    return out


def rust_structs():
    text = re.sub(r"//[^\n]*", "", open(LIB_RS).read())
    out = {}
    for name, body in re.findall(r"pub struct (\w+) \{(.*?)\}", text, flags=re.S):
        out[name] = [(f, " ".join(t.split())) for f, t in re.findall(r"pub (\w+):\s*([^,\n]+),", body)]
    return out


def rust_functions():
    text = re.sub(r"//[^\n]*", "", open(LIB_RS).read())
    out = {}
    for name, args, ret in re.findall(r"pub fn (rt_\w+)\(([^)]*)\)(?:\s*->\s*([^;]+))?;", text):
        params = [" ".join(a.split(":", 1)[1].split()) for a in args.split(",") if a.strip()]
        out[name] = (params, " ".join(ret.split()) if ret else None)
    return out


def test_functions_match():
    c, r = header_functions(), rust_functions()
    assert sorted(c) == sorted(r)
    for name in c:
        assert c[name] == r[name], name


def test_structs_match():
    c, r = header_structs(), rust_structs()
    assert sorted(c) == sorted(r)
    for name in c:
        assert c[name] == r[name], name


def test_opaque_handles_and_constants_match():
    htext = strip_c(open(HEADER).read())
    rtext = open(LIB_RS).read()
    for h in re.findall(r"typedef struct (\w+) \1;", htext):
        assert re.search(rf"pub enum {h} \{{\}}", rtext), h
    for name, val in re.findall(r"#define (RT_[A-Z_]+) \(?(-?\d+)u?\)?", htext):
        m = re.search(rf"pub const {name}: \w+ = (-?\d+);", rtext)
        assert m and int(m.group(1)) == int(val), name
