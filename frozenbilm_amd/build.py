"""Build libfbl.so (the gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m frozenbilm_amd.build            # incremental
    python -m frozenbilm_amd.build --force

hipcc cross-compiles for gfx950 without a GPU, so this also runs in the CPU-only build container; the resulting
``frozenbilm_amd/libfbl.so`` travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# FBL_DEBUG_BUILD=1 builds the measurement variant libfbl_dbg.so (-DFBL_DEBUG_SWITCHES: the FBL_* experiment switches of
# the dispatchers are live; frozenbilm_amd.lib loads it only when FBL_LIB points at it).  The product library has none.
DEBUG = os.environ.get("FBL_DEBUG_BUILD", "0") == "1"
OBJ = os.path.join(HERE, "csrc", "build_dbg" if DEBUG else "build")
LIB = os.path.join(HERE, "libfbl_dbg.so" if DEBUG else "libfbl.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]
if DEBUG:
    FLAGS.append("-DFBL_DEBUG_SWITCHES")
    FLAGS += os.environ.get("FBL_DEBUG_EXTRA_FLAGS", "").split()  # e.g. -DFBL_NO_SEED_DEV (A/B of code-generation effects)
    if os.environ.get("FBL_DEBUG_LIB_NAME"):
        LIB = os.path.join(HERE, os.environ["FBL_DEBUG_LIB_NAME"])
        OBJ = os.path.join(HERE, "csrc", "build_dbg_" + os.path.splitext(os.environ["FBL_DEBUG_LIB_NAME"])[0])


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 with gfx950 support)")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "fbl.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), _deps_mtime()):
        return obj
    cmd = [_hipcc(), *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    # A kernel that touches scratch (private) memory -- spilled registers, or a register array the compiler could not
    # keep in registers because a loop around it was not unrolled -- is several times slower than intended: refuse it.
    bad, name = [], "?"
    for line in r.stderr.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split("[")[0].strip()
        elif "ScratchSize [bytes/lane]:" in line:
            n = int(line.split("ScratchSize [bytes/lane]:")[1].split("[")[0])
            if n > 64:  # (a handful of spilled loop-invariant registers -- addresses -- is tolerated and only reported)
                bad.append(f"{name}: {n} bytes/lane")
            elif n > 0:
                print(f"[frozenbilm_amd.build] note: {name} spills {n} bytes/lane")
    if bad and os.environ.get("FBL_ALLOW_SCRATCH", "0") != "1":
        os.remove(obj)
        raise RuntimeError(f"{os.path.basename(src)}: kernels using scratch memory (set FBL_ALLOW_SCRATCH=1 to build anyway):\n  "
                           + "\n  ".join(bad))
    return obj


def build_lib(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[frozenbilm_amd.build] linked {LIB} from {len(objs)} objects")
    elif verbose:
        print(f"[frozenbilm_amd.build] {LIB} up to date")
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
