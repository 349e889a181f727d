#!/bin/bash
# usage: wsize.sh <lib>   -> conv time (bench) + WRITE_SIZE/FETCH_SIZE of k_conv3 for one step
R=$GRAFT_REPO_ROOT; L=$1
PBD_LIB=$R/$L python3 $R/bench.py --steps 5 --no-cpu-baseline --no-other-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['kernel_ms_per_step']['k_conv'])"
cd /tmp; export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
rm -rf $R/gpurun_out/pmc_x; PBD_LIB=$R/$L timeout -k 10 300 rocprofv3 --pmc $c --kernel-include-regex "k_conv3" --output-format csv -d $R/gpurun_out/pmc_x -- python3 $R/bench.py --steps 1 --warmup 0 --no-profile --no-cpu-baseline --no-other-mode > /dev/null 2>&1
python3 -c "
import csv,glob
t=0
for f in glob.glob('$R/gpurun_out/pmc_x/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)): t+=float(r['Counter_Value'])
print('   $c GB', round(t*1024/1e9,2))"
done
