#!/bin/bash
# same-box A/B of two builds on the FOCOPS update (configs[1] shape): bash tools/diag/focops_lib_ab.sh <other lib> [alternations]
N=${2:-3}
for i in $(seq 1 $N); do for L in "" "$1"; do
  export FSRL_HIP_LIB=$L; [ -z "$L" ] && unset FSRL_HIP_LIB
  FSRL_NO_CPU=1 FSRL_ONLY=focops python tools/bench_trust.py 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('${L:-new}', round(d['hip_ms_per_update'],3), 'ms', round(d['hip_us_per_step'],2), 'us per step')"
done; done
