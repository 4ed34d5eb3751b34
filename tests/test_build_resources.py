"""Compiler-reported resources of the kernels the sampling step actually launches: every kernel named in the latest
``profiles/r*_step_launches.txt`` (one replayed step of BASELINE configs[1], in launch order) must compile WITHOUT scratch -- a few
bytes per lane look harmless, but they sit inside persistent task loops -- and must keep the occupancy its workgroup size needs.
The figures come from ``hipcc -Rpass-analysis=kernel-resource-usage`` on the committed flags (protein_redesign_amd.build: recorded
while building, or re-derived here by a device-only compile; no GPU needed)."""
import glob
import os
import re

import pytest

from conftest import ROOT
from protein_redesign_amd import build


def step_kernels():
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_step_launches.txt")) if re.search(r"r\d+_step_launches\.txt$", f))
    assert files, "no profiles/r*_step_launches.txt"
    names = []
    for line in open(files[-1]):
        m = re.match(r"\s*\d+\s+[\d.]+ us\s+gap\s+[-\d.]+\s+grid\s+\d+\s+wg\s+(\d+)\s+(.*\S)\s*$", line)
        if m:
            names.append((m.group(2), int(m.group(1))))
    assert names, files[-1]
    return os.path.basename(files[-1]), sorted(set(names))


def lookup(usage, name):
    if name in usage:
        return [name]
    base = name.split("<")[0]
    args = name[len(base):].strip("<>")
    # a profile taken before a kernel gained a trailing template parameter names it by the leading ones: the default form is the one
    # whose added parameters have their first value (e.g. tri_attn_core_v3_kernel<64, 12> -> <64, 12, 0>, the round-3 key-loop form)
    more = sorted(k for k in usage if k.split("<")[0] == base and (not args or k[len(base):].strip("<>").startswith(args + ",")))
    return more[:1]


def test_default_dispatch_kernels_use_no_scratch():
    usage = build.resource_usage()
    assert len(usage) > 50
    src, kernels = step_kernels()
    missing, offenders = [], []
    for name, wg in kernels:
        hits = lookup(usage, name)
        if not hits:
            missing.append(name)
            continue
        for k in hits:
            u = usage[k]
            if u["scratch"]:
                offenders.append(f"{k}: {u['scratch']} B/lane scratch at {u['vgprs']} VGPRs")
            # the registers must leave room for the workgroup: waves per SIMD needed = wg / 256 (one workgroup per CU)
            need = -(-wg // 256)
            assert u["occupancy"] >= need, (k, u, wg)
    assert not missing, f"{src} names kernels the library no longer has (re-collect the profile): {missing}"
    assert not offenders, "scratch in kernels of the sampling step:\n  " + "\n  ".join(offenders)


def test_resource_table_is_complete():
    """every __global__ kernel of every source shows up (the parser keys on the demangled name)"""
    usage = build.resource_usage()
    for k in ("step_boundary_kernel<21>", "gemm_slab_kernel", "coord_head_kernel<64>"):
        assert lookup(usage, k), k
    assert all(u["vgprs"] is not None and u["scratch"] is not None and u["occupancy"] for u in usage.values())
