"""Every kernel launch of the LAST replayed step of a rocprofv3 --kernel-trace database of `bench.py` (hipGraph mode; steps are delimited by the
optimizer's adam_flat launches): start (ms since the step's first launch), duration (us), workgroups, short kernel name -- one line per launch, in
start order.  (Under the profiler the graph's parallel branches mostly run one after the other: durations are right, overlaps are not.)
    python tools/rocprof_step_list.py x_results.db > step_list.txt"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
names = {r[0]: r[1] for r in cur.execute(f"select id, display_name from {ks}")}
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
g = [c for c in ("grid_size_x", "grid_size_y", "workgroup_size_x") if c in cols]
rows = list(cur.execute(f"select kernel_id, start, end{''.join(', ' + c for c in g)} from {kd} order by start"))
adam = [i for i, r in enumerate(rows) if "adam_flat" in names[r[0]]]
lo, hi = adam[-4] + 1, adam[-1] + 1
t0 = rows[lo][1]
for r in rows[lo:hi]:
    n = names[r[0]]
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n).replace("void ", "")
    wg = ""
    if len(g) == 3:
        wg = f"{(r[3] // max(r[5], 1)) * max(r[4], 1):7d}"
    print(f"{(r[1] - t0) / 1e6:9.3f} {(r[2] - r[1]) / 1e3:9.1f} {wg} {n[:110]}")
