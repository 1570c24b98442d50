set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r5_gpu_all.log
tail -8 gpurun_out/r5_gpu_all.log
