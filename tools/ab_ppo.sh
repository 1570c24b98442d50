# same-box A/B of two builds on the headline bench:  bash tools/ab_ppo.sh <other lib> [alternations] [steps]
N=${2:-3}; K=${3:-10}
for i in $(seq 1 $N); do
for L in "" "$1"; do
  export FSRL_HIP_LIB=$L; [ -z "$L" ] && { unset FSRL_HIP_LIB; [ -n "${BASE_LIB:-}" ] && export FSRL_HIP_LIB=$BASE_LIB; }
  timeout 200 python bench.py --steps $K --warmup 3 --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${L:-new}', round(d['value'],2), round(d['roofline']['avg_launch_us'],2), round(d['roofline'].get('step_us',0),2))"
done; done
