"""Build libprd_hip.so in-tree with hipcc for gfx950 (no CMake / JIT cache: the .so travels with the repo)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libprd_hip.so")
SOURCES = ["prd_gemm.hip", "prd_pair.hip", "prd_tri.hip", "prd_tri2.hip", "prd_bwd.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
# per-source extras.  prd_tri2: the softmax arithmetic is placed by hand between the MFMAs of the key loop; the SLP vectoriser
# would pack its scalar fp32 adds into v_pk_add_f32 (slower beside MFMAs on gfx950) and move them out of their slots
EXTRA_FLAGS = {"prd_tri2.hip": ["-fno-slp-vectorize"]}


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
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", spath, "-o", obj]
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


def build_asan(verbose: bool = True) -> str:
    """Host-side AddressSanitizer build: libprd_hip_asan.so with the HOST code of every source instrumented
    (-fsanitize=address; device code is compiled as usual: -fno-gpu-sanitize, GPU ASan is not available on this pool) and the
    argument-validation driver tests/native/host_abi_check.c linked against it.  Runs on a machine without a GPU: every call of
    the driver is rejected by the argument checks before any HIP API is used.  Returns the path of the driver binary."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out_dir = os.path.join(HERE, "csrc", "asan")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(HERE, "libprd_hip_asan.so")
    objs = []
    for src in SOURCES:
        obj = os.path.join(out_dir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-g", "-fsanitize=address", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-fno-gpu-sanitize", "-o", lib] + objs)
    exe = os.path.join(out_dir, "host_abi_check")
    driver = os.path.join(os.path.dirname(HERE), "tests", "native", "host_abi_check.c")
    subprocess.check_call([hipcc, "-x", "c", driver, "-x", "none", "-g", "-fsanitize=address", "-fno-gpu-sanitize", "-o", exe, lib,
                           "-Wl,-rpath," + HERE])
    return exe


def build_timing(verbose: bool = True) -> str:
    """Diagnostic build with in-kernel cycle stamps (-DPRD_TIMING: tools/ta_timing.py, tools/phase_timing.py read them through
    prd_debug_read); load it with PRD_LIB=<path>.  Never the shipped library: the stamps cost ~10 % of a wave's cycles."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out_dir = os.path.join(HERE, "csrc", "timing")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(HERE, "libprd_hip_timing.so")
    objs = []
    for src in SOURCES:
        obj = os.path.join(out_dir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-DPRD_TIMING", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    if "--timing" in sys.argv:
        print(build_timing())
        sys.exit(0)
    if "--asan" in sys.argv:
        exe = build_asan()
        sys.exit(subprocess.call([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")))
    build(force="--force" in sys.argv)
