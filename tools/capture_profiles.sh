#!/bin/bash
# Round capture, run on the GPU box through gpurun:  bash tools/capture_profiles.sh r01
# Writes everything under gpurun_out/<tag>_*; tools/collect_profiles.py then copies the summaries to profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "import bench; print(bench.csrc_sha16())" > $O/${TAG}_csrc_sha16.txt
timeout 420 python bench.py --steps 20 --warmup 3 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_bench -- python $R/bench.py --steps 5 --warmup 2 --headline-only > $O/${TAG}_bench_profiled.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --headline-only > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python $R/bench.py --steps 2 --warmup 1 --headline-only > /dev/null 2>&1
# MFMA utilisation (north_star: "rocprof HBM GB/s and MFMA utilisation"): busy cycles of the matrix pipe, the fp32 MFMA
# op count and the GPU-active cycles per dispatch -- its own pass, kernel trace only
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmc_mfma -- python $R/bench.py --steps 2 --warmup 1 --headline-only > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmc_mfma_trust -- env FSRL_NO_CPU=1 python $R/tools/bench_trust.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmc_mfma_sac -- python $R/tools/bench_sac.py --rows 200000 --updates 200 --no-cpu > /dev/null 2>&1
# HBM-side traffic of the full-batch and replay updates: FETCH_SIZE / WRITE_SIZE in their own passes,
# one algorithm per run (bench_trust.py runs 1 warm-up + 5 timed updates; bench_sac.py 20 + 200)
for ALG in cpo trpo; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${TAG}_pmc_${C}_${ALG} -- env FSRL_NO_CPU=1 FSRL_ONLY=$ALG python $R/tools/bench_trust.py > /dev/null 2>&1
  done
done
# the same for CPO with round 5's split-K weight-gradient kernel (fsrl_tr_set_plan(wgrad = 2): the default up to r5)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${TAG}_pmc_${C}_cpo_splitk -- env FSRL_NO_CPU=1 FSRL_ONLY=cpo FSRL_TR_PLAN=0,0,2 python $R/tools/bench_trust.py > /dev/null 2>&1
done
# the same for CPO with the one-pass streaming weight-gradient kernel (fsrl_tr_set_plan(wgrad = 3): not the default)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${TAG}_pmc_${C}_cpo_stream -- env FSRL_NO_CPU=1 FSRL_ONLY=cpo FSRL_TR_PLAN=0,0,3 python $R/tools/bench_trust.py > /dev/null 2>&1
done
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${TAG}_pmc_${C}_sac -- python $R/tools/bench_sac.py --rows 200000 --updates 200 --no-cpu > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_trust -- env FSRL_NO_CPU=1 python $R/tools/bench_trust.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_group -- python $R/tools/bench_group.py --ks 8 --updates 3 > /dev/null 2>&1
cd $R
timeout 300 python tools/bench_group.py > $O/${TAG}_bench_group.json 2>/dev/null
timeout 300 python tools/bench_group.py --host-reset > $O/${TAG}_bench_group_hostreset.json 2>/dev/null
timeout 300 python tools/bench_group.py --no-clip > $O/${TAG}_bench_group_noclip.json 2>/dev/null
timeout 300 python tools/bench_sac.py > $O/${TAG}_bench_sac.json 2>/dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_sac -- python $R/tools/bench_sac.py --rows 200000 --updates 300 --no-cpu > /dev/null 2>&1
cd $R
timeout 300 python tools/bench_cvpo.py > $O/${TAG}_bench_cvpo.json 2>/dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_cvpo -- python $R/tools/bench_cvpo.py --updates 300 --no-cpu > /dev/null 2>&1
cd $R
timeout 600 python tools/bench_trust.py > $O/${TAG}_bench_trust.json 2>/dev/null
[ -z "${FSRL_CAPTURE_FAST:-}" ] && timeout 300 python tools/learning_curves.py > $O/${TAG}_learning_curves.json 2>/dev/null
if [ -z "${FSRL_CAPTURE_FAST:-}" ]; then       # FSRL_CAPTURE_FAST=1: skip what did not change this round (the r04 files stay the record)
timeout 300 python tools/learning_curves.py --task point-circle --agents ppol,cpo,sacl --epochs 25 > $O/${TAG}_learning_curves_pointcircle.json 2>/dev/null
timeout 300 python tools/bench_shmem.py > $O/${TAG}_bench_shmem.json 2>/dev/null
# microbenchmarks the DESIGN notes quote: device-wide barrier flavours, workgroup dispatch rate (prebuilt: tools/ubench/*.bin)
[ -x tools/ubench/gridsync.bin ] && { timeout 60 tools/ubench/gridsync.bin 217; timeout 60 tools/ubench/gridsync.bin 256; } > $O/${TAG}_ubench_gridsync.txt 2>&1
[ -x tools/ubench/dispatch.bin ] && timeout 60 tools/ubench/dispatch.bin > $O/${TAG}_ubench_dispatch.txt 2>&1
[ -x tools/ubench/chain.bin ] && timeout 120 tools/ubench/chain.bin 78 > $O/${TAG}_ubench_chain.txt 2>&1
fi
# r5: the GEMM inner loops in isolation (16x16x4 vs 32x32x2), where a 2-per-CU launch lands, the full-batch kernel plans side by side
[ -x tools/ubench/mfma_pat.bin ] && timeout 60 tools/ubench/mfma_pat.bin > $O/${TAG}_ubench_mfma_pat.txt 2>&1
[ -x tools/ubench/hwid.bin ] && timeout 60 tools/ubench/hwid.bin > $O/${TAG}_ubench_hwid.txt 2>&1
timeout 600 python tools/ab_trust_co.py --rounds 2 > $O/${TAG}_ab_trust_plans.json 2>/dev/null
# r6: the weight-gradient kernels against each other on the default tile plan (split-K | streaming | tile jobs: XCD-aware / plain / half the splits)
timeout 600 python tools/ab_trust_co.py --wgrad --rounds 2 > $O/${TAG}_ab_wgrad_plans.json 2>/dev/null
# per-kernel counter means of the trust-region run (co-resident kernels beside round 4's): summarised here, the raw files are large
cd /tmp
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $SET | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/${TAG}_pmcx/$T -- python $R/tools/ab_trust_co.py --rounds 1 --only cpo > /dev/null 2>&1
done
python $R/tools/pmc_summary.py $O/${TAG}_pmc_trust_plans.json /tmp/${TAG}_pmcx/* > /dev/null 2>&1
cd $R
ls $O | head -40
