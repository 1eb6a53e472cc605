"""Derived per-kernel figures from the one-counter-per-pass PMC tables of tools/pmc_step.sh (gpurun_out/pmc_step/<COUNTER>.csv):
    python tools/pmc_derive.py gpurun_out/pmc_step profiles/r3_pmc_summary.json profiles/r3_traffic.json [substring ...]
  MFMAs            = SQ_VALU_MFMA_BUSY_CYCLES / 32          (v_mfma_f32_32x32x16_bf16 occupies its SIMD's matrix pipe for 32 cycles)
  VALU per MFMA    = (SQ_INSTS_VALU - MFMAs) / MFMAs        (SQ_INSTS_VALU counts the MFMAs too)
  MFMA busy        = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)
  HBM bytes        = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024  (MI355X_MICROARCH.md: gfx950 FETCH_SIZE reports half of wide coalesced reads)"""
import csv
import json
import os
import sys

src, out_json, out_traffic = sys.argv[1], sys.argv[2], sys.argv[3]
subs = sys.argv[4:] or ["conv_halo_kernel", "wgrad_halo_kernel", "conv_bf16x6_kernel", "wgrad_bf16x6_kernel"]
tabs = {}
for f in os.listdir(src):
    if f.endswith(".csv"):
        with open(os.path.join(src, f)) as fh:
            tabs[f[:-4]] = {r["kernel"]: (int(r["dispatches"]), float(r["mean_KB_per_dispatch"])) for r in csv.DictReader(fh)}
kernels = sorted(k for k in tabs["SQ_VALU_MFMA_BUSY_CYCLES"] if any(s in k for s in subs)     # (stale tables of older passes may name other kernels)
                 and all(k in tabs.get(c, {}) for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE")))
res = {}
for k in kernels:
    g = lambda c: tabs.get(c, {}).get(k, (0, float("nan")))[1]
    short = k.split("::")[-1].split("(")[0]
    mf = g("SQ_VALU_MFMA_BUSY_CYCLES") / 32.0
    d = {"dispatches": tabs["SQ_VALU_MFMA_BUSY_CYCLES"].get(k, (0, 0))[0], "mfma_per_launch": round(mf), "insts_valu_per_launch": round(g("SQ_INSTS_VALU")),
         "non_mfma_valu_per_mfma": round((g("SQ_INSTS_VALU") - mf) / mf, 2) if mf else None,
         "lds_insts_per_mfma": round(g("SQ_INSTS_LDS") / mf, 2) if mf else None,
         "mfma_busy_frac": round(g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0 / (g("GRBM_GUI_ACTIVE") / 8.0), 3) if g("GRBM_GUI_ACTIVE") == g("GRBM_GUI_ACTIVE") else None,
         "fetch_kb_per_launch": g("FETCH_SIZE"), "write_kb_per_launch": g("WRITE_SIZE"),
         "hbm_bytes_per_launch": (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0}
    res[short] = d
json.dump({"note": __doc__, "kernels": res}, open(out_json, "w"), indent=1)
halo = [v for k, v in res.items() if k.startswith("conv_halo_kernel")]
n = sum(v["dispatches"] for v in halo)
tr = {"note": "launch-weighted mean over the conv_halo_kernel variants of the PMC passes in " + os.path.basename(out_json) + " (eager training step, B=8); bench.py reads kernels.conv_halo_kernel",
      "kernels": {"conv_halo_kernel": {"launches": n, "hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in halo) / max(n, 1)}}}
for k, v in res.items():
    tr["kernels"][k] = {"launches": v["dispatches"], "hbm_bytes_per_launch": v["hbm_bytes_per_launch"]}
json.dump(tr, open(out_traffic, "w"), indent=1)
for k, v in res.items():
    print(f"{k:55s} n={v['dispatches']:4d} MFMA/launch {v['mfma_per_launch']:>10} VALU/MFMA {v['non_mfma_valu_per_mfma']} LDS/MFMA {v['lds_insts_per_mfma']} busy {v['mfma_busy_frac']} HBM {v['hbm_bytes_per_launch'] / 1e6:.1f} MB")
