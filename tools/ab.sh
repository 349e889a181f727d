#!/bin/bash
# Same-box A/B of library builds (boxes differ by a few per cent, so two builds are only comparable inside one call):
#   tools/ab.sh <tag> lib1.so lib2.so ... [-- bench args]      -> gpurun_out/ab_<tag>.txt
# Each library runs the default bench workload twice, alternating, and the per-kernel milliseconds are printed.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" = "--" ] && shift
out=$R/gpurun_out/ab_$tag.txt; : > $out
for rep in 1 2; do
  for l in "${libs[@]}"; do
    PBD_LIB=$R/$l timeout -k 10 300 python3 $R/bench.py --steps 5 --no-cpu-baseline --no-other-mode "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']
print('%-52s rep $rep  %8.2f det/s  %7.3f ms/step  ' % ('$l', d['value'], d['ms_per_step']) + '  '.join('%s %.3f' % (n.replace('k_', ''), v) for n, v in k.items()))
" >> $out || exit 1
  done
done
cat $out
