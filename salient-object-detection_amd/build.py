"""Build libselfmask_hip.so in-tree with hipcc for gfx950 (no torch headers: the library is a plain C ABI)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libselfmask_hip.so")
SOURCES = ["gemm.hip", "gemm_f16x2.hip", "layernorm.hip", "attention.hip", "attention_f16x2.hip", "misc.hip", "eval.hip", "bilateral.hip", "forward.hip"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "common.h"),
                                                      os.path.join(HERE, "..", "include", "selfmask_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[selfmask_amd] " + " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
