"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite `*_results.db`) into the per-kernel stats table that
`rocprofv3 --stats` prints (name, calls, total ms, avg us, %).   python tools/rocprof_summary.py in.db out.csv"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, c, t, a, mn, mx in rows:
            w.writerow([n, c, int(t), f"{a:.1f}", f"{100.0 * t / tot:.2f}", int(mn), int(mx)])
    print(f"{len(rows)} kernels, total {tot/1e6:.2f} ms -> {out_path}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
