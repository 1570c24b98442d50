#!/bin/bash
# the headline step over several fresh processes: updates/s, the fused kernel's launch time and the step time (bench.py --headline-only),
# then the same under rocprofv3 for the per-kernel averages -- shows run-to-run modes of the three launches
N=${1:-6}
R=$PWD
for i in $(seq 1 $N); do
  python bench.py --steps 20 --warmup 3 --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i', round(d['value'],2), 'updates/s  fused', round(d['roofline']['avg_launch_us'],2), 'us  step', round(d['roofline'].get('step_us',0),2), 'us')"
done
cd /tmp; export TMPDIR=/tmp
for i in $(seq 1 3); do
  rm -rf /tmp/sm; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sm -o p -- python $R/bench.py --steps 5 --warmup 2 --headline-only > /dev/null 2>&1
  python - <<'PY'
import csv
rows = list(csv.DictReader(open('/tmp/sm/p_kernel_stats.csv')))
print('profiled', ' '.join(f"{r['Name'].split('(')[0].replace('void ','')[:24]}={float(r['AverageNs'])/1000:.2f}" for r in rows[:3]))
PY
done
