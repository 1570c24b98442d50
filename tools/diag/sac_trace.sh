cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sac
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sac -- python $GRAFT_REPO_ROOT/tools/bench_sac.py --no-cpu --updates 600 > /tmp/sac.log 2>&1
tail -2 /tmp/sac.log | cut -c1-400
python $GRAFT_REPO_ROOT/tools/trace_timeline.py /tmp/prof_sac 9
