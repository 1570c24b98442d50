export FSRL_HIP_LIB=fsrl_amd/libfsrl_hip_probe.so
for i in 1 2 3; do
for X in "" 1; do
  if [ -n "$X" ]; then export FSRL_WGRAD_XCD=1; else unset FSRL_WGRAD_XCD; fi
  timeout 150 python tools/bench_sac.py --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xcd=${X:-0}', 'sac us', round(1e3*d['ms_per_update'],2))"
done; done
