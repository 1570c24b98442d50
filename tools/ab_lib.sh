for i in 1 2; do
for L in "" tools/ubench/libfsrl_prev.bin; do
  export FSRL_HIP_LIB=$L
  [ -z "$L" ] && unset FSRL_HIP_LIB
  echo "== lib=${L:-new}"
  timeout 150 python tools/bench_cvpo.py --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cvpo us', round(d['us_per_update'],1))"
  timeout 150 python tools/bench_sac.py --no-cpu --rows 200000 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sac us', round(1e3*d['ms_per_update'],1))"
done; done
