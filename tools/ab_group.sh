#!/bin/bash
# same-box A/B of two builds of the library on the grouped PPO step (tools/bench_group.py, k = 4 and 8 members): the in-tree build against
# tools/ubench/libfsrl_prev.bin (a build of an earlier commit; git-ignored), alternated $1 times
N=${1:-3}
for i in $(seq 1 $N); do
for L in "" tools/ubench/libfsrl_prev.bin; do
  export FSRL_HIP_LIB=$L
  [ -z "$L" ] && unset FSRL_HIP_LIB
  timeout 200 python tools/bench_group.py --ks 4 8 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith('{'): continue
    d = json.loads(line)
    g = d.get('grouped') or d.get('results') or d
    print('lib=${L:-new}', json.dumps(g)[:400])
"
done; done
