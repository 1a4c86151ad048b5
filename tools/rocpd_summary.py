#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.
usage: tools/rocpd_summary.py <results.db> [> profiles/rNN_<name>_kernel_stats.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\[clone .*\]", "", name)
    return name if len(name) < 110 else name[:107] + "..."


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    print(f"# total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'calls':>7} {'total_ms':>11} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'share':>7}  kernel")
    for n, cnt, tot, avg, mn, mx in rows:
        print(f"{cnt:7d} {tot/1e6:11.3f} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*tot/total:6.2f}%  {short(n)}")


if __name__ == "__main__":
    main(sys.argv[1])
