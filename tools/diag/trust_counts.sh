cd /tmp && export TMPDIR=/tmp
for ALG in cpo trpo; do
rm -rf /tmp/prof_tc
FSRL_NO_CPU=1 FSRL_ONLY=$ALG rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tc -- python $GRAFT_REPO_ROOT/tools/bench_trust.py > /tmp/tc.log 2>&1
echo "== $ALG (1 warm-up + 5 timed updates)"
python $GRAFT_REPO_ROOT/tools/kstats.py $(find /tmp/prof_tc -name "*kernel_stats.csv") | head -12
done
