"""Kernel-time summary (name, calls, total/avg ms, %) from a rocprofv3 rocpd .db, written as CSV.

usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db profiles/rNN_kernel_stats.csv [steps]
"""
import csv
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end - start), min(end - start), max(end - start) from kernels group by {name_col}").fetchall()
    tot = sum(r[2] for r in rows)
    rows.sort(key=lambda r: -r[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"] + (["MsPerStep"] if steps else []))
        for n, c, t, mn, mx in rows:
            n = re.sub(r"\s+", " ", n)
            w.writerow([n, c, t, round(t / c, 1), round(100.0 * t / tot, 3), mn, mx] + ([round(t / steps / 1e6, 3)] if steps else []))
    print(f"{len(rows)} kernels, total {tot / 1e6:.1f} ms")
    for n, c, t, mn, mx in rows[:40]:
        print(f"{t / 1e6:9.2f} ms {100 * t / tot:5.1f}% {c:6d} x {t / c / 1e3:9.1f} us  {n[:110]}")


main()
