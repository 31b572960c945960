#!/usr/bin/env python3
"""Per-kernel statistics out of a rocprofv3 results database (rocpd SQLite: what `rocprofv3 --kernel-trace` writes by
default on ROCm 7.2) — the same table `--stats` prints, for runs whose CSV was not requested:

    python tools/prof_kernels.py gpurun_out/x/prof/pb_results.db [--top 30] [--csv out.csv]"""
import argparse
import csv
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--csv", default=None)
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "info_kernel_symbol" in t][0]
    rows = cur.execute(f"select s.kernel_name, count(*), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
                       f"sum(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name "
                       f"order by 6 desc").fetchall()
    total = sum(r[5] for r in rows) or 1
    out = [("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "TotalDurationNs", "Percentage")]
    for r in rows:
        out.append((r[0], r[1], round(r[2], 1), r[3], r[4], r[5], round(100.0 * r[5] / total, 2)))
    if a.csv:
        with open(a.csv, "w", newline="") as f:
            csv.writer(f).writerows(out)
    for r in out[:a.top + 1]:
        print("%-100s %6s %12s %14s %7s" % (str(r[0])[:100], r[1], r[2], r[5], r[6]))


if __name__ == "__main__":
    sys.exit(main())
