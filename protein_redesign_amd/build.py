"""Build libprd_hip.so in-tree with hipcc for gfx950 (no CMake / JIT cache: the .so travels with the repo)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libprd_hip.so")
SOURCES = ["prd_gemm.hip", "prd_pair.hip", "prd_tri.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "prd_common.h"), os.path.join(os.path.dirname(HERE), "include", "prd_hip.h")]
    objs = []
    for src in SOURCES:
        spath = os.path.join(CSRC, src)
        if not os.path.exists(spath):
            raise FileNotFoundError(f"HIP source listed in build.py is missing: {spath}")
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(obj, [spath] + headers):
            cmd = [hipcc] + FLAGS + ["-c", spath, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
