#!/bin/bash
# PMC counters for the DP kernels (batch 16, one step); output merged under gpurun_out/
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
run() { rm -rf $R/gpurun_out/$1; timeout -k 10 300 rocprofv3 --pmc $2 --kernel-include-regex "k_dt_|k_dp_combine" --output-format csv -d $R/gpurun_out/$1 -- python3 $R/bench.py --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-other-mode --conv-mode mfma > $R/gpurun_out/$1.log 2>&1; echo "$1 rc=$?"; }
run pmc_dt_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
run pmc_dt_b "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
run pmc_dt_c "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
run pmc_dt_d "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
