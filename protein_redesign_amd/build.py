"""Build libprd_hip.so in-tree with hipcc for gfx950 (no CMake / JIT cache: the .so travels with the repo)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libprd_hip.so")
SOURCES = ["prd_gemm.hip", "prd_pair.hip", "prd_tri.hip", "prd_tri2.hip", "prd_bwd.hip", "prd_spa.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
# per-source extras.  prd_tri2: the softmax arithmetic is placed by hand between the MFMAs of the key loop; the SLP vectoriser
# would pack its scalar fp32 adds into v_pk_add_f32 (slower beside MFMAs on gfx950) and move them out of their slots
EXTRA_FLAGS = {"prd_tri2.hip": ["-fno-slp-vectorize"]}


RESOURCE_JSON = os.path.join(CSRC, "resource_usage.json")     # per kernel: VGPRs, AGPRs, SGPRs, scratch bytes / lane, occupancy, LDS


def _demangle(names):
    """c++filt on the mangled kernel names; template arguments kept, the anonymous namespace and the parameter list dropped."""
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    except (OSError, subprocess.CalledProcessError):
        return list(names)
    res = []
    for n in out[:len(names)]:
        n = n.replace("(anonymous namespace)::", "")
        if n.startswith("void "):
            n = n[5:]
        depth, cut = 0, len(n)
        for i, ch in enumerate(n):          # the first '(' outside the template brackets starts the parameter list
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        res.append(n[:cut].strip())
    return res


def parse_resource_usage(stderr_text):
    """{kernel name: {vgprs, agprs, sgprs, scratch, occupancy, lds}} from hipcc -Rpass-analysis=kernel-resource-usage remarks."""
    import re
    blocks = re.split(r"remark: Function Name: ", stderr_text)[1:]
    mangled, rows = [], []
    for b in blocks:
        def g(key, b=b):
            m = re.search(key + r": (\d+)", b)
            return int(m.group(1)) if m else None
        mangled.append(b.split()[0])
        rows.append({"vgprs": g("VGPRs"), "agprs": g("AGPRs"), "sgprs": g("SGPRs"), "scratch": g(r"ScratchSize \[bytes/lane\]"),
                     "occupancy": g(r"Occupancy \[waves/SIMD\]"), "lds": g(r"LDS Size \[bytes/block\]")})
    return dict(zip(_demangle(mangled), rows))


def resource_usage(verbose: bool = False):
    """Register / scratch / occupancy figures of every kernel of the library, as the compiler reports them for the committed flags
    (hipcc cross-compiles without a GPU).  ``build()`` writes them next to the objects; a source whose figures are missing is
    analysed here (device code only, nothing linked).  tests/test_build_resources.py holds the default-dispatch kernels of the
    sampling step to ScratchSize == 0."""
    import json
    have = {}
    if os.path.exists(RESOURCE_JSON):
        try:
            with open(RESOURCE_JSON) as f:
                have = json.load(f)
        except (OSError, ValueError):
            have = {}
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "prd_common.h"), os.path.join(CSRC, "prd_tri2_v3_body.inc"), os.path.join(os.path.dirname(HERE), "include", "prd_hip.h")]
    changed = False
    for src in SOURCES:
        spath = os.path.join(CSRC, src)
        stamp = max(os.path.getmtime(d) for d in [spath] + headers)
        if src in have and have[src].get("stamp") == stamp:
            continue
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", spath, "-o", os.devnull]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"resource analysis of {src} failed:\n{r.stderr[-2000:]}")
        have[src] = {"stamp": stamp, "kernels": parse_resource_usage(r.stderr)}
        changed = True
    if changed:
        try:
            with open(RESOURCE_JSON, "w") as f:
                json.dump(have, f, indent=1, sort_keys=True)
        except OSError:
            pass
    return {k: v for src in SOURCES for k, v in have[src]["kernels"].items()}


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _record_resources(src, stderr_text, deps):
    import json
    have = {}
    if os.path.exists(RESOURCE_JSON):
        try:
            with open(RESOURCE_JSON) as f:
                have = json.load(f)
        except (OSError, ValueError):
            have = {}
    kernels = parse_resource_usage(stderr_text)
    if not kernels:
        return
    have[src] = {"stamp": max(os.path.getmtime(d) for d in deps), "kernels": kernels}
    try:
        with open(RESOURCE_JSON, "w") as f:
            json.dump(have, f, indent=1, sort_keys=True)
    except OSError:
        pass


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "prd_common.h"), os.path.join(CSRC, "prd_tri2_v3_body.inc"), os.path.join(os.path.dirname(HERE), "include", "prd_hip.h")]
    objs = []
    for src in SOURCES:
        spath = os.path.join(CSRC, src)
        if not os.path.exists(spath):
            raise FileNotFoundError(f"HIP source listed in build.py is missing: {spath}")
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(obj, [spath] + headers):
            # (-Rpass-analysis: the per-kernel resource remarks of THIS compilation are kept in csrc/resource_usage.json)
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-Rpass-analysis=kernel-resource-usage", "-fno-caret-diagnostics", "-c", spath, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stderr)
                raise subprocess.CalledProcessError(r.returncode, cmd)
            for line in r.stderr.splitlines():                      # warnings stay visible, the remarks do not
                if "remark:" not in line and line.strip():
                    sys.stderr.write(line + "\n")
            _record_resources(src, r.stderr, [spath] + headers)
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


def build_ab(verbose: bool = True) -> str:
    """libprd_hip_ab.so: the library with -DPRD_AB, i.e. INCLUDING the superseded kernels the shipped library leaves out (the
    first-generation split-16 attention cores of csrc/prd_tri.hip, in their three wave-count forms, and the fused form built on
    them).  For A/B measurements (PRD_LIB=<path> PRD_TA_VARIANT=10 ...) and for the parity tests of those kernels, which
    tests/test_ab_build.py runs against this library in a child process."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out_dir = os.path.join(HERE, "csrc", "ab")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(HERE, "libprd_hip_ab.so")
    headers = [os.path.join(CSRC, "prd_common.h"), os.path.join(CSRC, "prd_tri2_v3_body.inc"), os.path.join(os.path.dirname(HERE), "include", "prd_hip.h")]
    build(verbose=verbose)                                   # sources that never test PRD_AB share the shipped objects
    objs = []
    for src in SOURCES:
        spath = os.path.join(CSRC, src)
        with open(spath) as f:
            differs = "PRD_AB" in f.read()
        if not differs:
            objs.append(os.path.join(CSRC, src.replace(".hip", ".o")))
            continue
        obj = os.path.join(out_dir, src.replace(".hip", ".o"))
        if _stale(obj, [spath] + headers):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-DPRD_AB", "-c", spath, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if _stale(lib, objs):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


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
    if "--ab" in sys.argv:
        print(build_ab())
        sys.exit(0)
    if "--timing" in sys.argv:
        print(build_timing())
        sys.exit(0)
    if "--resources" in sys.argv:
        for name, u in sorted(resource_usage(verbose=True).items()):
            print(f"{u['vgprs']:4d} VGPR {u['agprs']:3d} AGPR {u['scratch']:5d} B scratch  occ {u['occupancy']}  LDS {u['lds']:6d}  {name}")
        sys.exit(0)
    if "--asan" in sys.argv:
        exe = build_asan()
        sys.exit(subprocess.call([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")))
    build(force="--force" in sys.argv)
