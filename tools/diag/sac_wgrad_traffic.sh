bash tools/ab_sac.sh 3
cd /tmp && export TMPDIR=/tmp
for P in 0 64; do
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_s
  FSRL_SAC_PLAN=$P rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_s -- python $GRAFT_REPO_ROOT/tools/bench_sac.py --rows 200000 --updates 200 --no-cpu > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob('/tmp/pmc_s/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0][:46]
        acc[n][0]+=float(r['Counter_Value']); acc[n][1]+=1
tot=0
for n,(v,c) in acc.items():
    if 'wgrad' in n: print('plan $P $C', n, round(v/c), 'per launch (raw units)', c)
PY
done; done
