#!/bin/bash
# how often does a fresh process land in the step's slow mode?  N processes of bench.py --headline-only: us per step of each
N=${1:-24}
for i in $(seq 1 $N); do
  python bench.py --steps 20 --warmup 3 --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline'].get('step_us',0),2), end=' ')"
done; echo
