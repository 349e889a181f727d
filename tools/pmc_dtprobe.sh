#!/bin/bash
# PMC counters of the distance-transform passes for one library build (timing probes of tools/build_variant.sh):
#   tools/pmc_dtprobe.sh <tag> <lib.so>      output: gpurun_out/pmc_dtprobe_<tag>.txt
tag=$1; lib=$2
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
export PBD_LIB=$R/$lib
out=$R/gpurun_out/pmc_dtprobe_$tag
rm -rf $out; mkdir -p $out
run() { timeout -k 10 300 rocprofv3 --pmc $2 --kernel-include-regex "k_dt_" --output-format csv -d $out/$1 -- python3 $R/bench.py --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-other-mode --conv-mode mfma > $out/$1.log 2>&1; echo "$1 rc=$?"; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" || exit 1
run b "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" || exit 1
python3 - $out << 'PY' | tee $out.txt
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][-44:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in agg.items():
    print(k)
    for n in sorted(c):
        print(f"   {n:28s} {c[n]:18.0f}")
PY
