#!/usr/bin/env python
"""Durations of every launch of ONE replayed step whose kernel name contains a substring, in launch order (rocprofv3
--kernel-trace database of bench.py).   usage: launch_list.py DB substring"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, start, end from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if "step_boundary" in r[0] or "reverse_update" in r[0]]
    a, b = idx[-2], idx[-1]
    out = [f"{(r[2] - r[1]) / 1e3:.1f}" for r in rows[a + 1: b + 1] if sys.argv[2] in r[0]]
    print(sys.argv[2], " ".join(out))


if __name__ == "__main__":
    main()
