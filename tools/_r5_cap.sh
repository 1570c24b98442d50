set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
FSRL_CAPTURE_FAST=1 bash tools/capture_profiles.sh r05 > gpurun_out/r05_capture.log 2>&1
du -sh gpurun_out; ls gpurun_out | head -60
# keep what comes back small: the collector needs the csv summaries only
find gpurun_out -name "*_kernel_trace.csv" -size +3M -delete
find gpurun_out -name "*agent_info.csv" -delete
du -sh gpurun_out
