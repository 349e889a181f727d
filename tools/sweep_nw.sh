#!/bin/bash
# usage: tools/sweep_nw.sh  (waves per workgroup of the exact convolution)
for n in 4 5 6; do
  PBD_CONV_NW=$n timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-mode 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nw', $n, d['value'], d['ms_per_step'], d['kernel_ms_per_step']['k_conv'])"
done
