#!/bin/bash
# Collects the rocprofv3 evidence of a round at the bench's default configuration:
#   kernel trace + stats, and HBM traffic counters (FETCH_SIZE and WRITE_SIZE in separate passes).
# usage (on the GPU box, via gpurun): tools/profile_round.sh <tag>
tag=${1:-r01}
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-ref > $R/gpurun_out/prof_$tag.log 2>&1; echo stats rc=$?
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-exact-ref > $R/gpurun_out/pmc_fetch_$tag.log 2>&1; echo fetch rc=$?
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-exact-ref > $R/gpurun_out/pmc_write_$tag.log 2>&1; echo write rc=$?
