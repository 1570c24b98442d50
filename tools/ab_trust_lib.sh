#!/bin/bash
# same-box A/B of two builds of the library on the full-batch updates (tools/bench_trust.py: CPO configs[2], TRPO-Lag on configs[1]'s shape):
# the in-tree build against tools/ubench/libfsrl_prev.bin (a build of an earlier commit; git-ignored), alternated $1 times
N=${1:-3}
for i in $(seq 1 $N); do
for L in "" tools/ubench/libfsrl_prev.bin; do
  export FSRL_HIP_LIB=$L
  [ -z "$L" ] && unset FSRL_HIP_LIB
  for ALG in cpo trpo; do
    FSRL_NO_CPU=1 FSRL_ONLY=$ALG timeout 200 python tools/bench_trust.py 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith('{'): continue
    d = json.loads(line)
    if 'hip_ms_per_update' in d: print('lib=${L:-new}', '$ALG', round(d['hip_ms_per_update'], 3), 'ms')
"
  done
done; done
