#!/bin/bash
# role timing of the replay agents' weight-gradient launches (fb_wgrad_kernel<.., NoRider | SgRider>): FSRL_WGRAD_SKIP on the probe build
# (bit 0 skips the dW2 tiles, bit 1 the aux blocks, bit 2 the db3 block; results invalid).  needs: bash fsrl_amd/csrc/build.sh --probes
R=$PWD; cd /tmp; export TMPDIR=/tmp
export FSRL_HIP_LIB=$R/fsrl_amd/libfsrl_hip_probe.so
for sk in 0 7 6 5 3; do
  rm -rf /tmp/ps; FSRL_WGRAD_SKIP=$sk rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o p -- python $R/tools/bench_sac.py --rows 200000 --updates 200 --no-cpu > /dev/null 2>&1
  python - "$sk" <<'PY'
import csv, sys
rows = list(csv.DictReader(open('/tmp/ps/p_kernel_stats.csv')))
print('skip', sys.argv[1], ' '.join(f"{r['Name'].split('(')[0].replace('void ','')[:34]}={float(r['AverageNs'])/1000:.1f}" for r in rows if 'wgrad' in r['Name']))
PY
done
