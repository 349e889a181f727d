#!/bin/bash
# PMC counters of the distance-transform pass kernels (batch 16, one step), aggregated per kernel name.
# usage (on the GPU box): tools/pmc_dtpass.sh <tag>      output: gpurun_out/pmc_dtpass_<tag>.txt
tag=${1:-x}
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/pmc_dtpass_$tag
rm -rf $out; mkdir -p $out
run() { timeout -k 10 300 rocprofv3 --pmc $2 --kernel-include-regex "k_dt_" --output-format csv -d $out/$1 -- python3 $R/bench.py --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-other-mode --conv-mode mfma > $out/$1.log 2>&1; echo "$1 rc=$?"; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
run b "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS"
run c "SQ_IFETCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64"
run d "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 GRBM_GUI_ACTIVE"
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
