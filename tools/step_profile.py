#!/usr/bin/env python
"""Per-kernel breakdown of ONE denoising step from a rocprofv3 --kernel-trace database of bench.py (the launches between the
last two reverse_update kernels).   usage: step_profile.py DB"""
import collections
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, start, end, grid_x from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if "reverse_update" in r[0] or "step_boundary" in r[0]]
    a, b = idx[-2], idx[-1]
    agg = collections.OrderedDict()
    for r in rows[a + 1: b + 1]:
        n = re.sub(r"\(.*", "", r[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))[:44]
        n += f" g{r[3]}"
        agg.setdefault(n, [0, 0.0])
        agg[n][0] += 1
        agg[n][1] += (r[2] - r[1]) / 1e3
    print(f"# one step: {(rows[b][2] - rows[a][2]) / 1e3:.1f} us, {b - a} launches")
    print(f"{'kernel (grid threads)':60s} {'n':>3s} {'total_us':>9s} {'avg_us':>8s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:60s} {c:3d} {t:9.1f} {t / c:8.1f}")


if __name__ == "__main__":
    main()
