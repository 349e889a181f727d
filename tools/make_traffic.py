"""profiles/traffic.json from the rocprofv3 PMC passes of tools/profile_round.sh.

HBM bytes per bench step and kernel = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KB;
on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section), so it
is doubled as that guide prescribes (uncalibrated for the narrower per-lane accesses of the DT passes; ratios
between variants are unaffected).  The passes ran exactly one bench step (--steps 1 --warmup 0)."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
names = {"k_resize4": "k_resize", "k_hog_tile": "k_hog_hist", "k_hog_grad4": "k_hog_hist", "k_hog_grad": "k_hog_hist", "k_conv_mfma<": "k_conv_mfma", "k_conv<": "k_conv", "k_conv3<": "k_conv", "k_dt_rows": "k_dt_rows", "k_dt_cols": "k_dt_cols", "k_dp_combine": "k_dp_combine",
         "k_hog_hist": "k_hog_hist", "k_hog_feat": "k_hog_feat", "k_resize": "k_resize", "k_pyrdown": "k_pyrdown", "k_dp_root": "k_dp_root"}
acc = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
import os
import shutil
for ctr, mode in (("fetch", "exact"), ("write", "exact"), ("fetch", "mfma"), ("write", "mfma")):
    f = max(glob.glob(f"gpurun_out/pmc_{ctr}_{tag}_{mode}/*/*counter_collection.csv"), key=os.path.getmtime)
    shutil.copy(f, f"profiles/{tag}_final_pmc_{ctr}_size_{mode}.csv")
    for r in csv.DictReader(open(f)):
        if mode == "mfma" and "k_conv_mfma" not in r["Kernel_Name"]:
            continue        # the other kernels are identical in both modes: take them from the exact run
        for pat, key in names.items():
            if (pat in r["Kernel_Name"]) if pat.endswith("<") else (pat + "<" in r["Kernel_Name"] or pat + "(" in r["Kernel_Name"]):
                acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
                if ctr == "fetch":
                    acc[key]["launches"] += 1
                break
out = {}
for k, v in acc.items():
    out[k] = {"fetch_size_kb": v["FETCH_SIZE"], "write_size_kb": v["WRITE_SIZE"], "launches_per_step": v["launches"],
              "hbm_bytes_per_step": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024}
json.dump(out, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"]):
    print(k, f"{v['hbm_bytes_per_step'] / 1e9:.2f} GB/step", v["launches_per_step"])

for mode in ("exact", "mfma"):
    f = max(glob.glob(f"gpurun_out/prof_{tag}_{mode}/*/*kernel_stats.csv"), key=os.path.getmtime)
    shutil.copy(f, f"profiles/{tag}_final_kernel_stats_{mode}.csv")
