set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_trust.py -x -q 2>&1 | tail -8 > gpurun_out/tail_test.log
timeout 900 python tools/ab_trust_co.py --rounds 2 > gpurun_out/tail_ab.log 2>&1
tail -4 gpurun_out/tail_test.log; tail -14 gpurun_out/tail_ab.log
