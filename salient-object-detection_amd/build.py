"""Build libselfmask_hip.so in-tree with hipcc for gfx950 (no torch headers: the library is a plain C ABI).

Each ``csrc/*.hip`` becomes an object under ``build/`` (recompiled only when it or a header changed, several in
parallel), then one link.  ``SM_TUNING=1`` (or ``build(tuning=True)``) defines ``SM_TUNING``: the timing-only ablation
switches of the GEMM / attention kernels exist only in that build (``lib/libselfmask_hip_tuning.so``), never in the
product library.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(LIB_DIR, "libselfmask_hip.so")
LIB_TUNING = os.path.join(LIB_DIR, "libselfmask_hip_tuning.so")
SOURCES = ["gemm.hip", "gemm_f16x2.hip", "gemm_w16.hip", "layernorm.hip", "attention.hip", "attention_f16x2.hip",
           "qkv_attention.hip", "misc.hip", "preprocess.hip", "eval.hip", "voting.hip", "cluster.hip", "spectral.hip", "bilateral.hip", "forward.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "selfmask_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build(tuning: bool = False) -> bool:
    lib = LIB_TUNING if tuning else LIB
    return _stale(lib, [os.path.join(CSRC, s) for s in _sources()] + HEADERS)


def build(force: bool = False, verbose: bool = True, tuning: bool = False, variant: str = "", defines=()) -> str:
    """``variant`` + ``defines``: an experiment build (A/B runs, scripts/ab_libs.sh): lib/libselfmask_hip_<variant>.so compiled
    with the given -D flags, objects under build/<variant>/; never loaded unless SM_HIP_LIB points at it."""
    tuning = tuning or os.environ.get("SM_TUNING") == "1"
    lib = LIB_TUNING if tuning else LIB
    if variant:
        lib, force = os.path.join(LIB_DIR, f"libselfmask_hip_{variant}.so"), True
    if not force and not needs_build(tuning):
        return lib
    os.makedirs(LIB_DIR, exist_ok=True)
    odir = os.path.join(OBJ_DIR, variant or ("tuning" if tuning else "product"))
    os.makedirs(odir, exist_ok=True)
    hipcc = _hipcc()
    extra = (["-DSM_TUNING=1"] if tuning else []) + [f"-D{d}" for d in defines]

    def compile_one(src):
        obj = os.path.join(odir, src.replace(".hip", ".o"))
        if force or _stale(obj, [os.path.join(CSRC, src)] + HEADERS):
            cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print("[selfmask_amd] " + " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=int(os.environ.get("SM_BUILD_JOBS", "6"))) as ex:
        objs = list(ex.map(compile_one, _sources()))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs
    if verbose:
        print("[selfmask_amd] " + " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return lib


if __name__ == "__main__":
    var = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--variant=")]
    print(build(force="--force" in sys.argv, tuning="--tuning" in sys.argv, variant=var[0] if var else "",
                defines=[a[2:] for a in sys.argv if a.startswith("-D")]))
