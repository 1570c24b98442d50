#!/bin/bash
# r6: per-kernel times of one algorithm's updates under two weight-gradient plans (rocprofv3 --kernel-trace --stats).
#   tools/prof_wgrad.sh <outdir> <alg: cpo|trpo> <plan> [<plan> ...]       plan = "tile_rows,hvp,wgrad" (fsrl_tr_set_plan)
set -u
OUT="$1"; ALG="$2"; shift 2
mkdir -p "$OUT"
export TMPDIR=/tmp
for PLAN in "$@"; do
    TAG="${ALG}_$(echo "$PLAN" | tr ',' '_')"
    rm -rf "/tmp/prof_$TAG"
    FSRL_TR_PLAN="$PLAN" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_$TAG" -o x -- \
        python tools/ab_trust_co.py --rounds 1 --only "$ALG" --plan "$PLAN" > "$OUT/$TAG.log" 2>&1
    F=$(find "/tmp/prof_$TAG" -name "*kernel_stats.csv" | head -1)
    [ -n "$F" ] && head -25 "$F" > "$OUT/${TAG}_kernel_stats.csv"
done
