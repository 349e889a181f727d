#!/bin/bash
# Collects the rocprofv3 evidence of a round (run on the GPU box via gpurun):
#   kernel trace + stats, and the HBM traffic counters (FETCH_SIZE and WRITE_SIZE in separate passes, one bench step each),
#   per convolution mode.
# usage: tools/profile_round.sh <tag> ["mode mode ..."] [bench args...]
#   tools/profile_round.sh r03                                   default workload (64 x 640x480), modes exact + mfma
#   tools/profile_round.sh r03_hd "exact mfma_f16" --rows 1080 --cols 1920 --batch 8
tag=${1:-r01}; shift
modes=${1:-"exact mfma"}; [ $# -gt 0 ] && shift
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
common="--no-cpu-baseline --no-other-mode $*"
for mode in $modes; do
  rm -rf $R/gpurun_out/prof_${tag}_$mode $R/gpurun_out/pmc_fetch_${tag}_$mode $R/gpurun_out/pmc_write_${tag}_$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_$mode -- python3 $R/bench.py --steps 3 --warmup 1 --conv-mode $mode $common > $R/gpurun_out/prof_${tag}_$mode.log 2>&1; echo "$mode stats rc=$?"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_${tag}_$mode -- python3 $R/bench.py --steps 1 --warmup 0 --no-profile --conv-mode $mode $common > $R/gpurun_out/pmc_fetch_${tag}_$mode.log 2>&1; echo "$mode fetch rc=$?"
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_${tag}_$mode -- python3 $R/bench.py --steps 1 --warmup 0 --no-profile --conv-mode $mode $common > $R/gpurun_out/pmc_write_${tag}_$mode.log 2>&1; echo "$mode write rc=$?"
done
