#!/bin/bash
# Same-box A/B of one build under different environments:   tools/ab_env.sh <tag> "ENV=a" "ENV=b" ... -- bench args
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "$1" = "--" ] && shift
out=$R/gpurun_out/abenv_$tag.txt; : > $out
for rep in 1 2; do
  for e in "${envs[@]}"; do
    env $e timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-other-mode "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']
print('%-24s rep $rep  %8.2f det/s  %7.3f ms/step  ' % ('$e', d['value'], d['ms_per_step']) + '  '.join('%s %.3f' % (n.replace('k_', ''), v) for n, v in k.items()))
" >> $out || exit 1
  done
done
cat $out
