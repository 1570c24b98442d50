set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
for rep in 1 2; do
  echo "== sc1 stores (product)"; timeout 600 python tools/ab_trust_co.py --rounds 1 2>&1 | tail -14 | grep "|"
  echo "== plain stores"; FSRL_HIP_LIB=$R/ab_libs/libfsrl_plainst.so timeout 600 python tools/ab_trust_co.py --rounds 1 2>&1 | tail -14 | grep "|"
done > gpurun_out/co4_stores.log 2>&1
cat gpurun_out/co4_stores.log
