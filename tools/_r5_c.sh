set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
tools/ubench/hwid.bin > gpurun_out/ubench_hwid.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "co_resident or plans_are_bit" 2>&1 | tail -30 > gpurun_out/co3_test.log
timeout 900 python tools/ab_trust_co.py --rounds 1 --delays > gpurun_out/co3_ab.log 2>&1
tail -5 gpurun_out/co3_test.log; tail -34 gpurun_out/co3_ab.log; cat gpurun_out/ubench_hwid.txt | tail -30
