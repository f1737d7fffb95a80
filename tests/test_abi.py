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
    assert handle.fbl_abi_version() == lib.ABI_VERSION == 8  # (lib.load refuses any other library: argument lists differ)
    assert handle.fbl_ln_bwd_ws_floats(1536) == 768 * 3 * 1536
    assert handle.fbl_colsum_ws_floats(100) == 512 * 100


def test_product_library_reads_no_environment(libpath):
    """The experiment switches (FBL_GEMM_*, FBL_ATTN_*) exist only in FBL_DEBUG_BUILD=1 builds: the product library
    does not even import getenv, and owns no stream (no hipStreamCreate*: helper streams are caller-provided)."""
    import subprocess

    syms = subprocess.run(["nm", "-D", "--undefined-only", libpath], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms
    assert "hipStreamCreate" not in syms
    # ... and every entry point is kernel launches (+ event fork / join onto the caller's aux stream) only: no memset / memcpy /
    # allocation calls.  A captured training step then consists of kernel nodes, ordered like eager launches -- the zero fills of the
    # accumulation targets were hipMemsetAsync for a while (memset nodes in the graphs) and replayed steps produced non-finite
    # gradients about once in 500 replays.
    for forbidden in ("hipMemset", "hipMemcpy", "hipMalloc", "hipFree", "hipHostMalloc", "hipDeviceSynchronize", "hipStreamSynchronize"):
        assert forbidden not in syms, forbidden


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


def test_adapter_bwd_dw_rejects_bad_arguments_on_the_host(libpath):
    """fbl_adapter_bwd_dw validates its tables before any launch (pure host logic, no GPU needed): group / segment limits,
    the bottleneck padding contract, stride alignment and ranges (include/fbl.h)."""
    from frozenbilm_amd import lib as L

    h = L.load(libpath)
    P = ctypes.c_void_p
    one = (P * 24)(*[0x1000] * 24)  # never dereferenced: every call below fails validation first

    def call(n=1, first=(0, 1), N=128, H=128, A=64, Ap=64, ld=(128, 64, 64, 128), dy=one, first_null=False):
        sf = None if first_null else (ctypes.c_int32 * len(first))(*first)
        lds = [(ctypes.c_int64 * max(n, 1))(*[v] * max(n, 1)) for v in ld]
        return h.fbl_adapter_bwd_dw(n, sf, dy, one, one, one, *lds, N, H, A, Ap, None, None, None, None)

    assert call(n=0) == 0 and call(N=0) == 0  # nothing to do
    assert call(n=17, first=tuple(range(18))) < 0  # more adapters than FBL_ADW_MAX_ADAPTERS
    assert call(first=(1, 2)) < 0          # seg_first[0] must be 0
    assert call(first=(0, 25)) < 0         # more segments than FBL_ADW_MAX_SEGMENTS
    assert call(first=(0, 0)) < 0          # an empty group
    assert call(first_null=True) < 0
    assert call(Ap=96) < 0 and call(Ap=320, A=300) < 0 and call(A=65, Ap=64) < 0  # Ap = A rounded up to 64, <= 256
    assert call(H=100) < 0                 # H % 8
    assert call(ld=(132, 64, 64, 128)) < 0  # stride alignment
    assert call(ld=(64, 64, 64, 128)) < 0   # ld_dy < H
    assert call(ld=(128, 32, 64, 128)) < 0  # ld_z < Ap
    assert call(dy=None) < 0
