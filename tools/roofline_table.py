#!/usr/bin/env python
"""Per-kernel roofline table of one denoising step: for every kernel of the replayed step

    launches / step, in-graph us per launch, algorithmic GFLOP and MB per launch, % of the fp32 MFMA peak the algorithmic flops
    reach, % of the 8 TB/s HBM peak the algorithmic bytes reach, MFMA-busy % (SQ_VALU_MFMA_BUSY_CYCLES), executed MFMA rate
    against the 2.5 PF 16-bit peak, VALU-active %, LDS bank-conflict share, HBM-side traffic (FETCH_SIZE x 2 + WRITE_SIZE)

usage: roofline_table.py TRACE_DB PMC_DB [PMC_DB ...] [--N 320 --P 64 --S 512 --b 1]
  TRACE_DB: rocprofv3 --kernel-trace of bench.py (graph replay: in-step durations); PMC_DBs: --pmc passes of bench.py --no-graph.
Every number can be recomputed from the printed columns and the formulas at the bottom of the output."""
import argparse
import collections
import re
import sqlite3

FP32_PEAK, PEAK16, HBM = 157.3e12, 2500e12, 8.0e12
NSIMD, NXCD = 1024, 8


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"[<(].*", "", name)


def algorithmic(N, P, S, H=4, c=16):
    """kernel -> (flops, bytes) per launch at b = 1 (U = N^2 P 4 bytes)."""
    U = N * N * P * 4
    Hc = H * c
    return {
        "tri_attn_core_v2_kernel": (8 * N * N * P * Hc + 4 * Hc * N ** 3, 2 * U),
        "tri_attn_core_v2l_kernel": (8 * N * N * P * Hc + 4 * Hc * N ** 3, 2 * U),
        "tri_attn_core_v3_kernel": (8 * N * N * P * Hc + 4 * Hc * N ** 3, 2 * U),
        "tri_attn_core_split_kernel": (8 * N * N * P * Hc + 4 * Hc * N ** 3, 2 * U),
        "tri_attn_core_split_long_kernel": (8 * N * N * P * Hc + 4 * Hc * N ** 3, 2 * U),
        "tri_attn_out_kernel": (2 * N * N * Hc * P, 3 * U),
        "pair_tail_h2_kernel": (2 * N * N * Hc * P + 16 * N * N * P * P + 2 * N * N * P * H, 3 * U),
        "tri_mul_proj_kernel": (8 * N * N * P * P, 3 * U),
        "tri_mul_contract_split_kernel": (2 * P * N ** 3, 3 * U),
        "tri_mul_out_kernel": (4 * N * N * P * P, 3 * U),
        "tri_mul_out_proj_kernel": (12 * N * N * P * P, 5 * U),
        "outer_linear_res_h2_kernel": (N * N * S * P + 4 * N * S * P, 2 * U),       # symmetric half of 2 N^2 S P
        "outer_linear_ks_kernel": (N * N * S * P + 4 * N * S * P, 2 * U),
        "pair_init_h2_kernel": (2 * N * N * 256 * P, 2 * U),
        # pair_init + OPM tail + the two first bias heads in one row pass: static pair in, pair out (+ 2 x [H,N,N] outputs)
        "pair_head_h2_kernel": (2 * N * N * 256 * P + 2 * N * N * (S // 4) * P + 2 * 2 * N * N * P * H, 2 * U + 2 * H * N * N * 4),
        "opm_pair_h2_kernel": (2 * N * N * (S // 4) * P, 2 * U),
        "coord_head_kernel": (2 * N * N * P * P + 2 * N * N * P, 2 * U),
        "pair_bias_kernel": (2 * 2 * N * N * P * H, U),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("pmc", nargs="*")
    ap.add_argument("--N", type=int, default=320)
    ap.add_argument("--P", type=int, default=64)
    ap.add_argument("--S", type=int, default=512)
    ap.add_argument("--b", type=int, default=1, help="complexes per GPU (bench.py --samples-per-gpu): algorithmic work per launch scales with it")
    a = ap.parse_args()
    alg = {k: (f * a.b, by * a.b) for k, (f, by) in algorithmic(a.N, a.P, a.S).items()}
    # in-graph durations: the launches between the last two step_boundary kernels
    db = sqlite3.connect(a.trace)
    rows = list(db.execute("select name, start, end from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if "step_boundary" in r[0] or "reverse_update" in r[0]]
    lo, hi = idx[-2], idx[-1]
    dur = collections.defaultdict(list)
    for r in rows[lo + 1: hi + 1]:
        dur[short(r[0])].append((r[2] - r[1]) / 1e3)
    step_us = (rows[hi][2] - rows[lo][2]) / 1e3
    # counters per launch (eager passes)
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in a.pmc:
        d = sqlite3.connect(path)
        for k, disp, cn, v in d.execute("select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection "
                                        "group by kernel_name, dispatch_id, counter_name"):
            cnt[short(k)][cn].append(v)

    def avg(k, c):
        v = cnt[k].get(c)
        return sum(v) / len(v) if v else None

    print(f"# one replayed step: {step_us:.1f} us, {hi - lo} launches (N = {a.N}, P = {a.P}, S = {a.S}, b = {a.b})")
    hdr = (f"{'kernel':32s} {'n':>3s} {'us':>7s} {'step%':>6s} {'GFLOP':>7s} {'MB':>7s} {'%fp32pk':>8s} {'%HBMpk':>7s} "
           f"{'MFMAbusy%':>9s} {'exec16 TF/s':>11s} {'%16bitpk':>8s} {'VALUact%':>8s} {'LDSconf%':>8s} {'traffic MB':>10s}")
    print(hdr)
    order = sorted(dur, key=lambda k: -sum(dur[k]))
    for k in order:
        n, us = len(dur[k]), sum(dur[k]) / len(dur[k])
        fl, by = alg.get(k, (None, None))
        cyc = avg(k, "GRBM_GUI_ACTIVE")
        cyc = cyc / NXCD if cyc else None
        busy = avg(k, "SQ_VALU_MFMA_BUSY_CYCLES")
        valu = avg(k, "SQ_ACTIVE_INST_VALU")
        conf, ldsact = avg(k, "SQ_LDS_BANK_CONFLICT"), avg(k, "SQ_LDS_IDX_ACTIVE")
        fetch, write = avg(k, "FETCH_SIZE"), avg(k, "WRITE_SIZE")
        f = lambda v, w=7, p=1: (f"{v:{w}.{p}f}" if v is not None else " " * (w - 1) + "-")      # noqa: E731
        mfma_busy = 100 * busy / (NSIMD * cyc) if busy and cyc else None
        # a 32x32x16 (16x16x32) 16-bit MFMA keeps the pipe busy 32 (16) cycles for 32768 (16384) flops: 1024 flops per busy cycle;
        # fp32 MFMA kernels (pair_bias, coord_head, single_attn_core, gemm_skinny) do 64 flops per busy cycle and are not priced here
        exec16 = busy * 1024 / (us * 1e-6) / 1e12 if busy else None
        print(f"{k[:32]:32s} {n:3d} {us:7.1f} {100 * n * us / step_us:6.1f} {f(fl / 1e9 if fl else None, 7, 2)} {f(by / 1e6 if by else None)} "
              f"{f(100 * fl / (us * 1e-6) / FP32_PEAK if fl else None, 8)} {f(100 * by / (us * 1e-6) / HBM if by else None)} "
              f"{f(mfma_busy, 9)} {f(exec16, 11)} {f(100 * exec16 * 1e12 / PEAK16 if exec16 else None, 8)} "
              f"{f(100 * 4 * valu / (NSIMD * cyc) if valu and cyc else None, 8)} {f(100 * conf / ldsact if conf is not None and ldsact else None, 8)} "
              f"{f((2 * fetch + write) * 1024 / 1e6 if fetch is not None and write is not None else None, 10)}")
    print("""
columns: us = average in-graph launch duration (rocprofv3 --kernel-trace of the replayed hipGraph); step% = n x us / step;
GFLOP, MB = ALGORITHMIC flops / bytes of one launch (SURVEY.md 8d, DESIGN.md 4.2; U = N^2 P 4 B); %fp32pk = GFLOP / us / 157.3 TF/s;
%HBMpk = MB / us / 8 TB/s; MFMAbusy% = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) of the eager launch;
exec16 TF/s = SQ_VALU_MFMA_BUSY_CYCLES x 1024 flops per busy cycle / in-graph us (what the 16-bit matrix pipe executed: the split
arithmetic issues 3 products per MAC), %16bitpk = that / 2500 TF/s; VALUact% = 4 x SQ_ACTIVE_INST_VALU / (1024 x cycles);
LDSconf% = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 wide-read correction).""")


if __name__ == "__main__":
    main()
