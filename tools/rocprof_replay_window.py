"""Steady-state per-step kernel table from a rocprofv3 --kernel-trace database of `bench.py` in hipGraph mode: only the
dispatches of the last N replayed steps (delimited by the optimizer's adam_flat launches, 3 per step) are aggregated.
    python tools/rocprof_replay_window.py gpurun_out/prof_replay/rp_results.db profiles/out.csv [steps=10]
Note: with the profiler attached the parallel branches of the graph (the two encoder passes) execute one after the other, so
the per-step SUM of kernel durations (what this table shows) is the serial work, not the step time."""
import collections
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
names = {r[0]: r[1] for r in cur.execute(f"select id, display_name from {ks}")}
rows = list(cur.execute(f"select kernel_id, start, end from {kd} order by start"))
adam = [r for r in rows if "adam_flat" in names[r[0]]]
t0 = adam[-3 * steps - 1][2]
sel = [r for r in rows if r[1] >= t0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    a = agg[names[r[0]]]
    a[0] += 1
    a[1] += (r[2] - r[1]) / 1e3
span = (sel[-1][2] - sel[0][1]) / 1e6
busy = sum(v[1] for v in agg.values()) / 1e3
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "CallsPerStep", "MsPerStep", "AverageUs", "Percentage"])
    for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
        w.writerow([n, round(c / steps, 1), round(t / 1e3 / steps, 3), round(t / c, 1), round(100 * t / 1e3 / busy, 2)])
print(f"{steps} steps: {span / steps:.2f} ms per step (profiled), sum of kernel durations {busy / steps:.2f} ms per step, "
      f"{len(sel) / steps:.0f} kernels per step -> {sys.argv[2]}")
