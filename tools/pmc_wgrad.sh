#!/bin/bash
# r6: counters of the weight-gradient kernels under several plans, one stream (tile_rows + 64) so that nothing co-runs.
#   tools/pmc_wgrad.sh <outdir> <alg> <plan> [<plan> ...]    -> <outdir>/pmc_<alg>_<plan>.json (per-kernel means)
# Each counter group is its own rocprofv3 pass (kernel trace only, as gpurun requires).
set -u
OUT="$1"; ALG="$2"; shift 2
mkdir -p "$OUT"; OUT="$(cd "$OUT" && pwd)"
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
CSETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum")
for PLAN in "$@"; do
    TAG="${ALG}_$(echo "$PLAN" | tr ',' '_')"
    DIRS=()
    i=0
    for G in "${CSETS[@]}"; do
        D="/tmp/pmc_${TAG}_$i"; i=$((i + 1))
        (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$D" -- \
            python "$R/tools/ab_trust_co.py" --rounds 1 --only "$ALG" --plan "$PLAN" > "$OUT/pmc_${TAG}_$i.log" 2>&1)
        DIRS+=("$D")
    done
    python "$R/tools/pmc_summary.py" "$OUT/pmc_$TAG.json" "${DIRS[@]}" > /dev/null
done
