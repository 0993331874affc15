"""Drift guard for the Rust FFI shim (integration/rust/goldibear-gpu/src/lib.rs), which cannot be compiled in this image:
every `extern "C"` declaration and every `#[repr(C)]` struct in it is parsed and compared with include/goldibear_gpu.h -
name, arity, each parameter's C type, the return type, and the structs' field names, order and widths.  The shim may bind
a subset of the header, but never something the header does not declare.  No GPU needed."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "goldibear_gpu.h")
SHIM = os.path.join(ROOT, "integration", "rust", "goldibear-gpu", "src", "lib.rs")

# canonical spelling of the types that cross the boundary
C_SCALARS = {"int": "i32", "int32_t": "i32", "int64_t": "i64", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "double": "f64",
             "char": "c_char", "void": "c_void", "gb_status": "i32", "uint8_t": "u8"}
OPAQUE = {"gb_ctx", "gb_batch", "gb_circuit", "gb_circuit_config", "gb_gate", "gb_challenger_state"}


def _strip_c(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return re.sub(r"^\s*#[^\n]*", "", src, flags=re.M)   # preprocessor lines (constants are read separately)


def _c_type(t):
    """'const gb_circuit_config*' -> ('gb_circuit_config', const=True, depth=1) in canonical spelling"""
    t = t.strip()
    depth = t.count("*")
    t = t.replace("*", " ")
    words = [w for w in t.split() if w not in ("struct",)]
    const = "const" in words
    words = [w for w in words if w != "const"]
    assert len(words) == 1, t
    base = words[0]
    base = C_SCALARS.get(base, base)
    assert base in set(C_SCALARS.values()) | OPAQUE, "unknown C type " + base
    return base, const and depth > 0, depth


def _rust_type(t):
    t = t.strip()
    depth, const = 0, False
    first = True
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+", t)
        assert m, t
        if first:
            const = m.group(1) == "const"   # constness of the outermost pointee, which is what the C declaration's const names
            first = False
        depth += 1
        t = t[m.end():]
    if t == "c_int":
        t = "i32"
    assert t in set(C_SCALARS.values()) | OPAQUE, "unknown Rust type " + t
    return t, const, depth


def _split_params(s):
    return [p.strip() for p in s.split(",") if p.strip() and p.strip() != "void"]


def header_functions():
    src = _strip_c(open(HEADER).read())
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w \t\*]*?)\b(gb_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        ps = []
        for p in _split_params(params):
            pm = re.match(r"(.*?)([A-Za-z_]\w*)$", p)
            ps.append((pm.group(2), _c_type(pm.group(1))))
        out[name] = (_c_type(ret), ps)
    return out


def header_structs():
    src = _strip_c(open(HEADER).read())
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ty, names = decl.split(None, 1)
            for nm in names.split(","):
                am = re.match(r"(\w+)\s*\[(\d+)\]$", nm.strip())   # `uint64_t sponge_state[16]` <-> `[u64; 16]`
                fields.append((am.group(1), "[%s; %s]" % (C_SCALARS[ty], am.group(2))) if am else (nm.strip(), C_SCALARS[ty]))
        out[m.group(3)] = fields
    return out


def shim_functions():
    src = re.sub(r"//[^\n]*", "", open(SHIM).read())
    out = {}
    for block in re.findall(r'extern\s+"C"\s*\{(.*?)\n\}', src, flags=re.S):
        for m in re.finditer(r"fn\s+(gb_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
            ps = []
            for p in _split_params(m.group(2)):
                nm, ty = p.split(":", 1)
                ps.append((nm.strip(), _rust_type(ty)))
            out[m.group(1)] = (_rust_type(m.group(3)) if m.group(3) else ("c_void", False, 0), ps)
    return out


def shim_structs():
    src = re.sub(r"//[^\n]*", "", open(SHIM).read())
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[[^\]]*\]\s*)*pub\s+struct\s+(\w+)\s*\{(.*?)\}", src, flags=re.S):
        fields = [(f.group(1), f.group(2).strip()) for f in re.finditer(r"pub\s+(\w+)\s*:\s*(\[[^\]]*\]|[^,\n]+)", m.group(2))]
        out[m.group(1)] = fields
    return out


def test_every_shim_declaration_matches_the_header():
    hdr, shim = header_functions(), shim_functions()
    assert len(hdr) >= 34 and len(shim) >= 20
    assert "gb_prove" in shim and "gb_commit_values" in shim and "gb_verify" in shim
    for name, (ret, params) in shim.items():
        assert name in hdr, "the shim binds %s, which the header does not declare" % name
        hret, hparams = hdr[name]
        assert ret == hret, "%s: return type %r, header %r" % (name, ret, hret)
        assert len(params) == len(hparams), "%s: %d parameters, header %d" % (name, len(params), len(hparams))
        for i, ((_, rt), (hn, ht)) in enumerate(zip(params, hparams)):
            assert rt[0] == ht[0] and rt[2] == ht[2], "%s: parameter %d (%s) is %r, header %r" % (name, i, hn, rt, ht)
            if ht[2] == 1:   # single pointers: `const T*` <-> `*const T`
                assert rt[1] == ht[1], "%s: parameter %d (%s) constness differs" % (name, i, hn)


def test_shim_takes_matrices_as_separately_allocated_columns():
    """the reference's witness is Vec<Vec<F>> and from_values takes Vec<PolynomialValues<F>> (iop/witness.rs:277-279,
    fri/oracle.rs:68-75): the shim binds the column-pointer entry points with a `*const *const c_void` table and its safe wrappers
    build that table from the columns' own pointers - no flattening copy (flat_map / concat / extend_from_slice) anywhere"""
    shim = shim_functions()
    for name in ("gb_commit_values_cols", "gb_commit_coeffs_cols", "gb_prove_cols", "gb_prove_retry_cols", "gb_prove_salted_cols",
                 "gb_zs_partial_products_cols", "gb_host_alloc", "gb_host_register", "gb_ctx_set_option"):
        assert name in shim, name
    assert shim["gb_prove_cols"][1][1][1] == ("c_void", True, 2) and shim["gb_commit_values_cols"][1][2][1] == ("c_void", True, 2)
    src = open(SHIM).read()
    body = src[src.index("fn column_table"):]
    assert "as_ref().as_ptr()" in body
    for wrapper, call in (("fn prove_columns", "gb_prove_cols("), ("fn prove_retry_columns", "gb_prove_retry_cols("),
                          ("fn commit_columns", "gb_commit_values_cols"), ("fn zs_partial_products_columns", "gb_zs_partial_products_cols(")):
        assert wrapper in src and call in src[src.index(wrapper):src.index(wrapper) + 2500], wrapper
    code = re.sub(r"//[^\n]*", "", src)
    for flatten in ("flat_map", ".concat()", "extend_from_slice", ".flatten()"):
        assert flatten not in code, "the shim flattens a matrix with " + flatten
    assert "GB_INPUT_P3_REPR" in src and "P3InMemory" in src


def test_repr_c_structs_match_field_for_field():
    hdr, shim = header_structs(), shim_structs()
    for name in ("gb_circuit_config", "gb_gate", "gb_challenger_state"):
        assert name in hdr and name in shim
        assert shim[name] == hdr[name], "%s differs:\n shim   %r\n header %r" % (name, shim[name], hdr[name])
    # the opaque handles carry no fields
    for name in ("gb_ctx", "gb_batch", "gb_circuit"):
        assert [t for _, t in shim[name]] in ([], ["[u8; 0]"]) or all(n.startswith("_") for n, _ in shim[name])


def test_shim_constants_match_the_header():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    consts = {k: int(v) for k, v in re.findall(r"\b(GB_[A-Z0-9_]+)\s*=\s*(\d+)", src)}
    consts.update({k: int(v) for k, v in re.findall(r"#define\s+(GB_[A-Z0-9_]+)\s+(\d+)\b", src)})
    rust = {k: int(v) for k, v in re.findall(r"pub const (GB_[A-Z0-9_]+): [iu]\d+ = (\d+);", open(SHIM).read())}
    assert rust, "no constants found in the shim"
    for k, v in rust.items():
        assert consts.get(k) == v, "%s = %d in the shim, %r in the header" % (k, v, consts.get(k))


def test_the_parser_sees_drift():
    """the guard itself: a changed arity / type / field order must be reported"""
    hdr = header_functions()
    ret, params = hdr["gb_prove"]
    assert [p[0] for p in params] == ["c", "witness", "flags", "public_inputs", "num_public_inputs", "proof_out", "proof_cap",
                                      "proof_len"]
    assert params[3][1] == ("u64", True, 1) and params[7][1] == ("usize", False, 1) and ret == ("i32", False, 0)
    assert _rust_type("*mut *mut gb_ctx") == ("gb_ctx", False, 2) and _rust_type("*const c_void") == ("c_void", True, 1)
    assert header_structs()["gb_circuit_config"][-1] == ("num_public_inputs", "u32")
