#!/bin/bash
# quick per-kernel rocprofv3 stats of one bench configuration (run on the GPU box): tools/kstats.sh <conv-mode> <tag>
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kstats_$2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstats_$2 -- python3 $R/bench.py --steps 3 --warmup 1 --conv-mode $1 --no-cpu-baseline --no-other-mode > $R/gpurun_out/kstats_$2.log 2>&1
f=$(find $R/gpurun_out/kstats_$2 -name '*kernel_stats.csv' | head -1)
python3 - "$f" << 'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
