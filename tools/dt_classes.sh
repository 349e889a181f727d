#!/bin/bash
# Per-class timing of the distance-transform passes (run on the GPU box): one bench step under rocprofv3 --kernel-trace,
# then the k_dt_pass dispatches aggregated by (pass, dynamic LDS size).  usage: tools/dt_classes.sh <tag> [bench args]
tag=$1; shift
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/dtc_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/dtc_$tag -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-mode --no-profile "$@" > $R/gpurun_out/dtc_$tag.log 2>&1
f=$(find $R/gpurun_out/dtc_$tag -name '*kernel_trace.csv' | head -1)
python3 - "$f" << 'PY' | tee $R/gpurun_out/dtc_$tag.txt
import csv, sys, collections
agg = collections.OrderedDict()
rows = list(csv.DictReader(open(sys.argv[1])))
half = len(rows) // 2           # warm-up step first, timed step second: keep the second half
for r in rows[half:]:
    n = r["Kernel_Name"]
    if "k_dt_pass" not in n and "k_dt_rows" not in n and "k_dt_cols" not in n and "k_dp_combine" not in n:
        continue
    key = (n.split("(")[0][-40:], int(r["LDS_Block_Size"]), int(r["VGPR_Count"]))
    a = agg.setdefault(key, [0, 0.0, 0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; a[2] += int(r["Grid_Size_X"]) // 64
tot = 0.0
for (n, lds, vg), (c, ms, waves) in agg.items():
    tot += ms
    print(f"{n:42s} lds {lds:7d} vgpr {vg:4d} launches {c:4d} waves {waves:9d} total_ms {ms:8.3f} us/kwave {1e6*ms/max(waves,1):9.1f}")
print("sum ms", round(tot, 3))
PY
