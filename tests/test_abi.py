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


def test_product_library_reads_no_environment(libpath):
    """The experiment switches (FBL_GEMM_*, FBL_ATTN_*) exist only in FBL_DEBUG_BUILD=1 builds: the product library
    does not even import getenv, and owns no stream (no hipStreamCreate*: helper streams are caller-provided)."""
    import subprocess

    syms = subprocess.run(["nm", "-D", "--undefined-only", libpath], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms
    assert "hipStreamCreate" not in syms


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


def test_gemm_plan_host_query(libpath):
    """fbl_gemm_plan is pure host logic (no launch, no GPU): the shapes of the step that the dispatcher gives to the
    8-phase kernel -- what bench.py's `roofline` attributes to the dominant kernel -- and the ones it does not."""
    from frozenbilm_amd import lib as L

    for shp in [(8512, 6144, 1536), (9024, 4608, 1536), (8512, 1536, 6144), (8512, 1728, 6144), (8512, 1536, 1536),
                (8512, 1536, 4608), (8512, 128100, 1536)]:
        assert L.gemm_plan(*shp) == 8, shp
    for shp in [(8512, 1536, 192), (8512, 192, 1536), (391, 64, 10240), (4100, 3584, 128), (512, 1536, 3072),
                (8512, 1536, 1600)]:
        assert L.gemm_plan(*shp) == 2, shp
    assert L.gemm_plan(8512, 6144, 1536, batch=2) == 2 and L.gemm_plan(8512, 6144, 1536, splitk=4) == 2
