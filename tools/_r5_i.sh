set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
  echo "== product"; timeout 600 python tools/ab_trust_co.py --rounds 1 2>&1 | grep "default\|full R-op (0,4)"
  echo "== setprio"; FSRL_HIP_LIB=$R/ab_libs/libfsrl_setprio.so timeout 600 python tools/ab_trust_co.py --rounds 1 2>&1 | grep "default\|full R-op (0,4)"
done > gpurun_out/co5_setprio.log 2>&1
cat gpurun_out/co5_setprio.log
