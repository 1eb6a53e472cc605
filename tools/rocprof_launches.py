"""Per-launch durations (us) of the kernels whose name contains any of the given substrings, over the LAST replayed step of a
rocprofv3 --kernel-trace database of `bench.py` in hipGraph mode (steps are delimited by the optimizer's adam_flat launches).
    python tools/rocprof_launches.py gpurun_out/prof/x_results.db fewout grid_sample_bwd
    python tools/rocprof_launches.py x_results.db --chains 30 65536     # every launch >= 30 us with <= 65536 threads: few workgroups
                                                                        # running long dependent chains (latency-bound whatever the size)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pats = sys.argv[2:]
chains = None
if pats and pats[0] == "--chains":
    chains = (float(pats[1]), int(pats[2]))
    pats = []
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
names = {r[0]: r[1] for r in cur.execute(f"select id, display_name from {ks}")}
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
gcols = [c for c in ("grid_size_x", "workgroup_size_x") if c in cols]
rows = list(cur.execute(f"select kernel_id, start, end{''.join(', ' + c for c in gcols)} from {kd} order by start"))
adam = [r for r in rows if "adam_flat" in names[r[0]]]
t0 = adam[-4][2]
for r in rows:
    if r[1] < t0:
        continue
    n = names[r[0]]
    if chains is not None:
        ok = (r[2] - r[1]) / 1e3 >= chains[0] and len(r) > 3 and r[3] <= chains[1]
    else:
        ok = any(p in n for p in pats)
    if ok:
        short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        print(f"{(r[1] - t0) / 1e6:9.3f} ms  {short:48s} {(r[2] - r[1]) / 1e3:9.1f} us  grid {r[3] if len(r) > 3 else ''}")
