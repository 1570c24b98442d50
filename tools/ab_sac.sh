#!/bin/bash
# same-box A/B of two builds of the library on the SAC-Lag update of BASELINE configs[3] (tools/bench_sac.py, 1 M-row store):
# the in-tree build against tools/ubench/libfsrl_prev.bin (a build of the previous commit; git-ignored), alternated $1 times
N=${1:-3}
for i in $(seq 1 $N); do
for L in "" tools/ubench/libfsrl_prev.bin; do
  export FSRL_HIP_LIB=$L
  [ -z "$L" ] && unset FSRL_HIP_LIB
  timeout 150 python tools/bench_sac.py --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=${L:-new}', 'sac us', round(1e3*d['ms_per_update'],2))"
done; done
