#!/bin/bash
# PMC counters of the matrix-core convolution (batch 16, one step). usage (GPU box): tools/pmc_mfma.sh <tag> <mfma|mfma_f16>
tag=${1:-x}; mode=${2:-mfma}
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/pmc_mfma_$tag
rm -rf $out; mkdir -p $out
run() { timeout -k 10 300 rocprofv3 --pmc $2 --kernel-include-regex "k_conv_mfma" --output-format csv -d $out/$1 -- python3 $R/bench.py --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-other-mode --conv-mode $mode > $out/$1.log 2>&1; echo "$1 rc=$?"; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
run b "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS"
run c "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"
python3 - $out << 'PY' | tee $out.txt
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][-44:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in agg.items():
    print(k)
    for n in sorted(c):
        print(f"   {n:30s} {c[n]:18.0f}")
PY
