#!/usr/bin/env python
"""Summarise a rocprofv3 results database (rocpd sqlite, `--kernel-trace --stats`) as text:
per-kernel calls / total / average / share, like the CSV stats view.   usage: rocprof_summary.py DB [OUT]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    return name[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1].split('/')[-1]} (durations in us)",
             f"{'kernel':70s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}"]
    for name, calls, total, avg, pct in rows:
        lines.append(f"{short(name):70s} {calls:7d} {total:12.1f} {avg:10.2f} {pct:7.2f}")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
