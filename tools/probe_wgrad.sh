#!/bin/bash
# tile-only / aux-only timing of fb_wgrad_kernel (probe build): FSRL_WGRAD_SKIP bit 1 = skip tile blocks, 2 = skip aux blocks, 4 = skip db3
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for SK in 0 1 2; do
  rm -rf /tmp/pw_$SK
  FSRL_WGRAD_SKIP=$SK FSRL_HIP_LIB=$R/fsrl_amd/libfsrl_hip_probe.so FSRL_NO_CPU=1 FSRL_ONLY=${ALG:-cpo} timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw_$SK -- python $R/tools/bench_trust.py > /dev/null 2>&1
  f=$(ls /tmp/pw_$SK/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "skip=$SK"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'fb_wgrad' in r['Name'] or 'hvp_mixed' in r['Name'] or 'tile_mixed' in r['Name']: print('  ', r['Name'][:45], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1))"
done
