#!/bin/bash
# timing experiment: fb_wgrad_kernel with roles skipped (results invalid)
# needs the probe build:  bash fsrl_amd/csrc/build.sh --probes
R=$PWD; cd /tmp; export TMPDIR=/tmp
export FSRL_HIP_LIB=$R/fsrl_amd/libfsrl_hip_probe.so
for sk in 0 1 2 4 3 7; do
  rm -rf /tmp/pw; FSRL_ONLY=${FSRL_ONLY:-} FSRL_NO_CPU=1 FSRL_WGRAD_SKIP=$sk rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -o p -- python $R/tools/bench_trust.py > /dev/null 2>&1
  python - "$sk" <<'PY'
import csv, sys
rows = list(csv.DictReader(open('/tmp/pw/p_kernel_stats.csv')))
print('skip', sys.argv[1], ' '.join(f"{r['Name'][5:32]}={float(r['AverageNs'])/1000:.1f}us" for r in rows if 'wgrad' in r['Name']))
PY
done
