#!/bin/bash
# same-box A/B of the read-back plans of the trust-region updates: polled completion words (default) against hipMemcpyAsync +
# hipStreamSynchronize (fsrl_tr_set_plan tile_rows + 128); CPO configs[2] and TRPO-Lag, $1 alternations
N=${1:-3}
for i in $(seq 1 $N); do for P in 0,0,0 128,0,0; do
  for A in cpo trpo; do
    FSRL_TR_PLAN=$P FSRL_NO_CPU=1 FSRL_ONLY=$A python tools/bench_trust.py 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('plan $P', d['bench'], round(d['hip_ms_per_update'],3))"
  done
done; done
