set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_trust.py -x -q 2>&1 | tail -30 > gpurun_out/gn_test.log
timeout 900 python tools/ab_trust_co.py --rounds 2 > gpurun_out/gn_ab.log 2>&1
tail -8 gpurun_out/gn_test.log; tail -12 gpurun_out/gn_ab.log
