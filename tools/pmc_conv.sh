#!/bin/bash
# PMC counters for the exact convolution kernel (batch 16, one step); output merged under gpurun_out/
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
run() { rm -rf $R/gpurun_out/$1; timeout -k 10 300 rocprofv3 --pmc $2 --kernel-include-regex "k_conv" --output-format csv -d $R/gpurun_out/$1 -- python3 $R/bench.py --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-other-mode > $R/gpurun_out/$1.log 2>&1; echo "$1 rc=$?"; }
run pmc_conv_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
run pmc_conv_b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"
run pmc_conv_c "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_LEVEL_WAVES SQ_CYCLES"
python3 - $R/gpurun_out << 'PY' | tee $R/gpurun_out/pmc_conv.txt
import csv, glob, sys, collections
agg = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/pmc_conv_[abc]/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
for n in sorted(agg):
    print(f"   {n:28s} {agg[n]:18.0f}")
PY
