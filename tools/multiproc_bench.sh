# N independent bench.py processes sharing ONE GPU (how FSRL users run several seeds): aggregate updates/s
N=${1:-3}
for i in $(seq 1 $N); do
  ( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])" > /tmp/mp_$i.txt ) &
done
wait
python - <<PY
import glob
v=[float(open(f).read()) for f in sorted(glob.glob('/tmp/mp_*.txt'))[:$N]]
print("processes", len(v), "each", [round(x,1) for x in v], "aggregate", round(sum(v),1))
PY
rm -f /tmp/mp_*.txt
