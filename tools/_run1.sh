cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fullsize.py -q -x -k "plans" 2>&1 | tail -15
python -m pytest tests/test_gpu_trust.py -q -x 2>&1 | tail -5
for plan in 16,1 0,2 0,0; do
  echo "plan $plan"; FSRL_TR_PLAN=$plan FSRL_NO_CPU=1 FSRL_ONLY=cpo python tools/bench_trust.py
  FSRL_TR_PLAN=$plan FSRL_NO_CPU=1 FSRL_ONLY=trpo python tools/bench_trust.py
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03a_prof_cpo -- env FSRL_NO_CPU=1 FSRL_ONLY=cpo python $GRAFT_REPO_ROOT/tools/bench_trust.py > /dev/null 2>&1
