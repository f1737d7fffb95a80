"""CPU: the C-ABI shared library builds, loads and exports exactly what include/fbl.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "fbl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|int64_t)\s+(fbl_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(3).strip()
        n = 0 if args in ("", "void") else args.count(",") + 1
        out[m.group(2)] = (m.group(1), n)
    return out


@pytest.fixture(scope="module")
def libpath():
    from frozenbilm_amd.build import build_lib

    return build_lib(verbose=False)


def test_header_and_binding_agree(libpath):
    from frozenbilm_amd import lib

    decl = declared_functions()
    assert len(decl) >= 28
    assert set(decl) == set(lib.SIGNATURES), set(decl) ^ set(lib.SIGNATURES)
    for name, (ret, nargs) in decl.items():
        res, argtypes = lib.SIGNATURES[name]
        assert len(argtypes) == nargs, name
        assert (res is ctypes.c_int64) == (ret == "int64_t"), name


def test_library_exports_every_symbol(libpath):
    h = ctypes.CDLL(libpath)
    for name in declared_functions():
        assert hasattr(h, name), name
    from frozenbilm_amd import lib

    handle = lib.load(libpath)
    assert handle.fbl_abi_version() == 1
    assert handle.fbl_ln_bwd_ws_floats(1536) == 768 * 3 * 1536
    assert handle.fbl_colsum_ws_floats(100) == 512 * 100


def test_missing_library_fails_loudly(tmp_path):
    from frozenbilm_amd import lib

    with pytest.raises(RuntimeError, match="no CPU/eager fallback"):
        lib.load(str(tmp_path / "libfbl.so"))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "frozenbilm_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)
