#!/bin/bash
# usage: tools/ab_libs.sh <conv-mode> lib1.so lib2.so ...   (A/B whole-library variants with bench.py)
mode=$1; shift
for l in "$@"; do
  PBD_LIB=$PWD/partsbaseddetector_amd/$l timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --conv-mode $mode 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$l', d['value'], d['ms_per_step'], {x:k[x] for x in ('k_conv','k_hog_hist','k_dt_rows','k_dt_cols','k_dp_combine')})"
done
