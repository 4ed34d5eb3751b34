#!/usr/bin/env python
"""Every launch of ONE replayed step in launch order: index, duration (us), grid threads, workgroup size, kernel name
(rocprofv3 --kernel-trace database of bench.py).   usage: step_launches.py DB"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    wg = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
    q = f"select name, start, end, grid_x{', ' + wg if wg else ''} from kernels order by start"
    rows = list(db.execute(q))
    idx = [i for i, r in enumerate(rows) if "reverse_update" in r[0] or "step_boundary" in r[0]]
    a, b = idx[-2], idx[-1]
    print(f"# one step: {(rows[b][2] - rows[a][2]) / 1e3:.1f} us, {b - a} launches")
    for k, r in enumerate(rows[a + 1: b + 1]):
        n = re.sub(r"\(.*", "", r[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))[:60]
        gap = (r[1] - rows[a + k][2]) / 1e3
        print(f"{k:3d} {(r[2] - r[1]) / 1e3:8.1f} us  gap {gap:5.1f}  grid {r[3]:7d}" + (f" wg {r[4]:5d}" if wg else "") + f"  {n}")


if __name__ == "__main__":
    main()
