#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counters from one or more rocpd databases.
usage: pmc_summary.py DB [DB ...]   (prints kernel, launches, avg duration us, avg of every counter)"""
import collections
import re
import sqlite3
import sys


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", name)[:44]


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        rows = db.execute("select kernel_name, dispatch_id, counter_name, sum(value), max(duration) from counters_collection "
                          "group by kernel_name, dispatch_id, counter_name")
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(dict)
        for k, d, cn, v, du in rows:
            per[short(k)][cn].append(v)
            dur[short(k)][d] = du
        names = sorted({cn for k in per for cn in per[k]})
        print(f"# {path}")
        print(f"{'kernel':44s} {'n':>5s} {'avg_us':>9s} " + " ".join(f"{n[:22]:>22s}" for n in names))
        order = sorted(per, key=lambda k: -sum(dur[k].values()))
        for k in order:
            n = len(dur[k])
            line = f"{k:44s} {n:5d} {sum(dur[k].values()) / n / 1e3:9.1f} "
            line += " ".join(f"{sum(per[k][cn]) / max(1, len(per[k][cn])):22.4g}" for cn in names)
            print(line)


if __name__ == "__main__":
    main()
