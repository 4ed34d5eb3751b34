#!/usr/bin/env python
"""Static instruction mix of the loops of every kernel in a gfx950 .s file (hipcc --save-temps).

On gfx950 the fp32 MFMAs and the VALU share one SIMD's issue time (tools/ubench/coissue_bench.hip: cycles add,
with any number of waves), so  mfma_cycles / (mfma_cycles + valu_cycles)  of a hot loop is its ceiling as a
fraction of the fp32 MFMA peak.  Cost model (measured): v_mfma 32x32x2 = 64 cycles, 16x16x4 = 32, plain VALU
~4.6, transcendental ~9.2.   usage: isa_mix.py file.s [kernel-substring]"""
import re
import sys
from collections import Counter

TRANS = ("v_exp_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_log_f32", "v_sin_f32", "v_cos_f32")


def cost(op):
    if op.startswith("v_mfma_f32_32x32x2"):
        return "mfma", 64.0
    if op.startswith("v_mfma_f32_16x16x4"):
        return "mfma", 32.0
    if op.startswith("v_mfma"):
        return "mfma", 32.0
    if op.startswith(TRANS):
        return "valu", 9.2
    if op.startswith("v_"):
        return "valu", 4.6
    if op.startswith(("ds_", "global_", "buffer_", "scratch_", "flat_")):
        return "mem", 0.0
    return "salu", 0.0


def main():
    text = open(sys.argv[1]).read().split("\n")
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    kernels, cur, name = {}, None, None
    for line in text:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
        elif cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                cur = None
    for name, lines in kernels.items():
        if pat not in name:
            continue
        labels = {}
        for i, l in enumerate(lines):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = i
        loops = []
        for i, l in enumerate(lines):
            m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        print(f"== {name[:100]}")
        for a, b in sorted(loops):
            body = [x.strip().split()[0] for x in lines[a:b + 1]
                    if x.strip() and not x.strip().startswith((";", ".")) and not x.strip().endswith(":")]
            agg, cnt = Counter(), Counter()
            for op in body:
                k, c = cost(op)
                agg[k] += c
                cnt[k] += 1
            if cnt["mfma"] == 0:
                continue
            frac = agg["mfma"] / (agg["mfma"] + agg["valu"])
            top = Counter(op for op in body if op.startswith("v_") and not op.startswith("v_mfma")).most_common(8)
            print(f"  loop @{a}-{b}: {len(body)} instr, mfma {cnt['mfma']} ({agg['mfma']:.0f} cyc), valu {cnt['valu']} "
                  f"({agg['valu']:.0f} cyc), mem {cnt['mem']}, salu {cnt['salu']}  -> ceiling {100 * frac:.0f}%")
            print("      " + ", ".join(f"{o} {n}" for o, n in top))


if __name__ == "__main__":
    main()
