#!/bin/bash
# Collects the rocprofv3 evidence of a round at the bench's default configuration (run on the GPU box via gpurun):
#   kernel trace + stats of the default (exact) run and of the fast (MFMA) run,
#   HBM traffic counters (FETCH_SIZE and WRITE_SIZE in separate passes, one bench step each) for both modes.
# usage: tools/profile_round.sh <tag>
tag=${1:-r01}
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
common="--no-cpu-baseline --no-other-mode"
for mode in exact mfma; do
  rm -rf $R/gpurun_out/prof_${tag}_$mode $R/gpurun_out/pmc_fetch_${tag}_$mode $R/gpurun_out/pmc_write_${tag}_$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_$mode -- python3 $R/bench.py --steps 3 --warmup 1 --conv-mode $mode $common > $R/gpurun_out/prof_${tag}_$mode.log 2>&1; echo "$mode stats rc=$?"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_${tag}_$mode -- python3 $R/bench.py --steps 1 --warmup 0 --no-profile --conv-mode $mode $common > $R/gpurun_out/pmc_fetch_${tag}_$mode.log 2>&1; echo "$mode fetch rc=$?"
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_${tag}_$mode -- python3 $R/bench.py --steps 1 --warmup 0 --no-profile --conv-mode $mode $common > $R/gpurun_out/pmc_write_${tag}_$mode.log 2>&1; echo "$mode write rc=$?"
done
