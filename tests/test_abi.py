"""CPU-only: the C-ABI shared library builds, loads and exports exactly what include/recalgo.h
declares; the ctypes binding covers every declaration.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "recalgo.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(recalgo_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    from recalgorithm_amd import build
    return build.build(verbose=False)


def test_header_declares_functions():
    fns = declared_functions()
    assert "recalgo_embedding_gather_fwd" in fns and "recalgo_cross_fwd" in fns
    assert len(fns) >= 20


def test_library_exports_every_declared_symbol(lib_path):
    import torch  # noqa: F401  maps the HIP runtime first
    lib = ctypes.CDLL(lib_path)
    missing = [f for f in declared_functions() if not hasattr(lib, f)]
    assert not missing, f"declared in recalgo.h but not exported: {missing}"


def test_ctypes_binding_matches_header(lib_path):
    from recalgorithm_amd import _lib
    declared = set(declared_functions())
    bound = set(_lib.SIGNATURES)
    assert declared == bound, f"header-only: {declared - bound}; binding-only: {bound - declared}"
    lib = _lib.load()
    assert lib.recalgo_abi_version() == _lib.ABI_VERSION
    m = re.search(r"#define RECALGO_ABI_VERSION (\d+)", open(HEADER).read())
    assert m and int(m.group(1)) == _lib.ABI_VERSION
    assert lib.recalgo_target_arch() == b"gfx950"


def declaration_hash(path=HEADER):
    """sha256 over the header's declarations: comments, the version number and white space removed."""
    import hashlib
    src = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    src = re.sub(r"#define RECALGO_ABI_VERSION \d+", "", src)
    return hashlib.sha256(re.sub(r"\s+", " ", src).strip().encode()).hexdigest()


def test_declarations_do_not_change_without_a_version_bump():
    """include/recalgo.abi: one `version sha256` line per ABI version.  A stale librecalgo_hip.so with re-ordered
    arguments corrupts calls silently; _lib.load() catches it only if the version moved with the declarations."""
    version = int(re.search(r"#define RECALGO_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    recorded = dict((int(v), h) for v, h in (ln.split() for ln in open(os.path.join(ROOT, "include", "recalgo.abi"))
                                             if ln.strip() and not ln.startswith("#")))
    h = declaration_hash()
    assert version == max(recorded), f"recalgo.h is at ABI {version}, include/recalgo.abi ends at {max(recorded)}"
    assert recorded[version] == h, (
        f"the declarations of include/recalgo.h changed (sha256 {h}) but RECALGO_ABI_VERSION is still {version}: bump it in "
        f"recalgo.h and _lib.py and append `<version> {h}` to include/recalgo.abi")
    assert len(set(recorded.values())) == len(recorded), "two ABI versions with identical declarations"


def test_header_arg_counts_match_binding():
    from recalgorithm_amd import _lib
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^)]*)\)", src)
        assert m, name
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), f"{name}: header has {len(params)} params, binding {len(args)}"


def test_header_arg_types_match_binding():
    """Every parameter and return type of recalgo.h against the ctypes signature (int vs int64_t vs float vs
    pointer): a wrong width here silently corrupts arguments at call time."""
    from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p
    from recalgorithm_amd import _lib
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)

    def ctype_of(decl):
        decl = decl.strip()
        if "*" in decl or decl.startswith("recalgo_stream_t"):
            return c_void_p
        base = decl.rsplit(" ", 1)[0].replace("const ", "").strip()
        return {"int": c_int, "int64_t": c_int64, "float": c_float, "double": c_double, "unsigned": c_int, "unsigned int": c_int}[base]

    for name, (res, args) in _lib.SIGNATURES.items():
        m = re.search(r"([A-Za-z_0-9 \*]+?)\b" + name + r"\s*\(([^)]*)\)", src)
        assert m, name
        ret = m.group(1).strip()
        want_res = c_char_p if "char" in ret else {"int": c_int, "int64_t": c_int64}[ret.replace("const ", "")]
        assert res is want_res, f"{name}: returns {ret}, binding {res}"
        params = [p for p in m.group(2).split(",") if p.strip() and p.strip() != "void"]
        for i, (decl, bound) in enumerate(zip(params, args)):
            assert ctype_of(decl) is bound, f"{name} arg {i} `{decl.strip()}`: binding {bound.__name__}"


def test_object_code_is_gfx950(lib_path):
    data = open(lib_path, "rb").read()
    assert b"gfx950" in data


def test_missing_library_fails_loudly(tmp_path):
    from recalgorithm_amd import _lib
    saved = _lib._lib
    _lib._lib = None
    try:
        with pytest.raises(_lib.RecalgoError):
            _lib.load(str(tmp_path / "nope.so"))
    finally:
        _lib._lib = saved


def test_product_path_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
    (a child process running `-m oracle.cpu_baseline`) may touch it.  Static scan of the product sources."""
    import ast
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = []
    for path in glob.glob(os.path.join(root, "recalgorithm_amd", "**", "*.py"), recursive=True) + [os.path.join(root, "bench.py")]:
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            mods = []
            if isinstance(node, ast.Import):
                mods = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom) and node.level == 0 and node.module:
                mods = [node.module]
            if any(m == "oracle" or m.startswith("oracle.") for m in mods):
                offenders.append(f"{os.path.relpath(path, root)}:{node.lineno}")
    assert not offenders, offenders
    # bench.py reaches the oracle only through the cpu_baseline child process
    src = open(os.path.join(root, "bench.py")).read()
    assert '"-m", "oracle.cpu_baseline"' in src
