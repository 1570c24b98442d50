set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_trust.py tests/test_gpu_loop.py tests/test_gpu_facade.py -x -q 2>&1 | tail -12 > gpurun_out/ov_test.log
timeout 900 python tools/ab_trust_co.py --rounds 2 > gpurun_out/ov_ab.log 2>&1
tail -6 gpurun_out/ov_test.log; tail -12 gpurun_out/ov_ab.log
