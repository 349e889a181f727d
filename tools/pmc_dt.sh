#!/bin/bash
# PMC counters for the DP kernels (batch 16, one step); output merged under gpurun_out/
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS \
  --kernel-include-regex "k_dt_|k_dp_combine" --output-format csv -d $R/gpurun_out/pmc_dt3 -- python3 $R/bench.py --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --conv-mode mfma > $R/gpurun_out/pmc_dt3.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE \
  --kernel-include-regex "k_dt_|k_dp_combine" --output-format csv -d $R/gpurun_out/pmc_dt4 -- python3 $R/bench.py --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --conv-mode mfma > $R/gpurun_out/pmc_dt4.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ \
  --kernel-include-regex "k_dt_|k_dp_combine" --output-format csv -d $R/gpurun_out/pmc_dt5 -- python3 $R/bench.py --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --conv-mode mfma > $R/gpurun_out/pmc_dt5.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE \
  --kernel-include-regex "k_dt_|k_dp_combine" --output-format csv -d $R/gpurun_out/pmc_dt6 -- python3 $R/bench.py --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --conv-mode mfma > $R/gpurun_out/pmc_dt6.log 2>&1
echo done
