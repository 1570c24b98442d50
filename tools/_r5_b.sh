set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "co_resident or plans_are_bit" 2>&1 | tail -30 > gpurun_out/co2_test.log
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/co2_prof -- python $R/tools/ab_trust_co.py --rounds 2 > $R/gpurun_out/co2_ab.log 2> $R/gpurun_out/co2_ab.err
cp $(find /tmp/co2_prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/co2_kernel_stats.csv
timeout 600 python $R/tools/ab_trust_co.py --rounds 0 --splits > $R/gpurun_out/co2_splits.log 2>&1
cd $R
tail -5 gpurun_out/co2_test.log; tail -14 gpurun_out/co2_ab.log; tail -22 gpurun_out/co2_splits.log; head -8 gpurun_out/co2_kernel_stats.csv | cut -c1-150
