cd /tmp && export TMPDIR=/tmp
for K in 4 8; do
rm -rf /tmp/prof_g
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -- python $GRAFT_REPO_ROOT/tools/bench_group.py --ks $K --updates 3 > /tmp/g.log 2>&1
echo "== k=$K"; tail -1 /tmp/g.log | cut -c1-300
python $GRAFT_REPO_ROOT/tools/kstats.py $(find /tmp/prof_g -name "*kernel_stats.csv") | head -5
done
