"""The -DPRD_AB library (protein_redesign_amd.build.build_ab -> libprd_hip_ab.so): the superseded first-generation split-16 attention
cores are compiled out of the shipped libprd_hip.so and keep their parity tests HERE, in a child process that loads
the A/B library through PRD_LIB: the fused previous-update form, and -- with PRD_TA_VARIANT=10, the first-generation dispatch -- the
operator-level triangle-attention tests against the oracle."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_ab_library_exports_the_same_abi():
    """(no GPU) every entry point of the header is exported by the A/B library too; it is built on demand"""
    from protein_redesign_amd import build
    lib = build.build_ab(verbose=False)
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    have = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("prd_")}
    ship = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "protein_redesign_amd", "libprd_hip.so")],
                          capture_output=True, text=True, check=True).stdout
    want = {ln.split()[-1] for ln in ship.splitlines() if " T " in ln and ln.split()[-1].startswith("prd_")}
    assert have == want and len(have) > 50


@pytest.mark.gpu
def test_first_generation_kernels_in_the_ab_library():
    from protein_redesign_amd import build
    lib = build.build_ab(verbose=False)
    env = dict(os.environ, PRD_LIB=lib, PRD_TA_VARIANT="10")
    env.pop("PRD_LDS_POISON", None)
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_parity.py"), "-m", "gpu", "-q", "-x",
           "-k", "fused_previous_update or test_triangle_attention[ or large_logit_spread", "-p", "no:cacheprovider"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], tail


@pytest.mark.gpu
def test_superseded_second_generation_forms_in_the_ab_library():
    """Round 6: the shipped library carries ONE form of tri_attn_core_v3 / _v2l per pair_dim; key-loop forms 1-3, the round-3 phase 1 and the
    next-row prefetch (measured in rounds 3-5, none faster) compile only with -DPRD_AB and are checked here against the default form."""
    from protein_redesign_amd import build
    lib = build.build_ab(verbose=False)
    env = dict(os.environ, PRD_LIB=lib, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("PRD_LDS_POISON", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ab_forms_check.py")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ab forms ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
