"""rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv  ->  per-kernel mean counter value per dispatch.
    python tools/pmc_per_kernel.py <dir>/p_counter_collection.csv out.csv"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches", "mean_KB_per_dispatch"])
    for k, (n, s) in sorted(agg.items(), key=lambda x: -x[1][1]):
        w.writerow([k, n, round(s / n, 1)])
print(len(agg), "kernels ->", sys.argv[2])
