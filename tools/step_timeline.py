"""Timeline of ONE replayed training step from a rocprofv3 --kernel-trace database of bench.py (hipGraph mode): per 1-ms bucket the kernel
time by category (sum over concurrent streams, so > 1 ms per bucket means overlap) -- shows which phase of the step is made of what.
    python tools/step_timeline.py <rp_results.db> [step_from_end=2]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
names = {r[0]: r[1] for r in cur.execute(f"select id, display_name from {ks}")}
rows = list(cur.execute(f"select kernel_id, start, end from {kd} order by start"))
adam = [i for i, r in enumerate(rows) if "adam_flat" in names[r[0]]]
# a step = dispatches between the last adam_flat of step k-1 and the last adam_flat of step k (3 adam_flat launches per step)
e = adam[-1 - 3 * (back - 1)]
s = adam[-1 - 3 * back] + 1
sel = rows[s:e + 1]


def cat(n):
    for key, c in (("conv_halo", "halo conv"), ("wgrad_halo", "halo wgrad"), ("conv_bf16x6", "row conv"), ("wgrad_bf16x6", "row wgrad"),
                   ("conv_small", "small"), ("wgrad_small", "small"), ("conv_mfma", "fp32 tiles"), ("wgrad_mfma", "fp32 tiles"), ("bn_", "bn"),
                   ("attention", "attn/ln"), ("layernorm", "attn/ln"), ("gelu", "attn/ln"), ("grid_sample", "sampler"), ("resize", "sampler"),
                   ("corr_lookup", "sampler"), ("fewout", "fewout"), ("pack", "pack/adam"), ("adam", "pack/adam"), ("absmax", "pack/adam"),
                   ("at::", "aten"), ("Cijk", "aten")):
        if key in n:
            return c
    return "other"


t0 = sel[0][1]
span = (sel[-1][2] - t0) / 1e6
nb = int(span) + 1
buckets = [collections.defaultdict(float) for _ in range(nb)]
for kid, st, en in sel:
    c = cat(names[kid])
    a, b = (st - t0) / 1e6, (en - t0) / 1e6
    i = int(a)
    while a < b and i < nb:
        hi = min(b, i + 1)
        buckets[i][c] += hi - a
        a = hi
        i += 1
cats = sorted({c for b in buckets for c in b})
print(f"step span {span:.2f} ms, {len(sel)} kernels")
print("ms   " + " ".join(f"{c[:9]:>9s}" for c in cats) + "   total")
for i, b in enumerate(buckets):
    print(f"{i:3d}  " + " ".join(f"{b.get(c, 0):9.2f}" for c in cats) + f"  {sum(b.values()):6.2f}")
