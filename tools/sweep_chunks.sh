#!/bin/bash
# usage: tools/sweep_chunks.sh <conv-mode>
for c in 1 2 4 8; do
  PBD_PIPELINE_CHUNKS=$c timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --conv-mode $1 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunks', $c, d['value'], d['ms_per_step'])"
done
