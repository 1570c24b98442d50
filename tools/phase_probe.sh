#!/bin/bash
# timing experiment: kernel averages with the debug early-exit phases (results invalid)
# needs the probe build:  bash fsrl_amd/csrc/build.sh --probes   (the shipped library has no probes)
R=$PWD; cd /tmp; export TMPDIR=/tmp
export FSRL_HIP_LIB=$R/fsrl_amd/libfsrl_hip_probe.so
for ph in ${PHASES:-0 9 1 4 5 6 7 20 21 22 23}; do
  rm -rf /tmp/pp; FSRL_DBG_PHASE=$ph rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python $R/bench.py --steps 2 --warmup 1 ${BENCH_FLAGS:---no-cpu-baseline} > /dev/null 2>&1
  python - "$ph" <<'PY'
import csv, sys
rows = list(csv.DictReader(open('/tmp/pp/p_kernel_stats.csv')))
out = []
for r in rows:
    n = r['Name']
    for key in ('ppo_fwd_bwd', 'ppo_wgrad', 'adam_clip', 'prepare_pass', 'mlp_infer'):
        if key in n:
            out.append(f"{key}={float(r['AverageNs'])/1000:.1f}us")
print('phase', sys.argv[1], ' '.join(sorted(out)))
PY
done
