#!/bin/bash
# Build libfsrl_hip.so for gfx950 (cross-compiles without a GPU).  In-tree output so the
# .so travels to the GPU box with the repo snapshot.
#   build.sh            -> fsrl_amd/libfsrl_hip.so        (the product: no timing probes, no env switches)
#   build.sh --probes   -> fsrl_amd/libfsrl_hip_probe.so  (-DFSRL_PROBES: early-exit phase probes for
#                          tools/phase_probe.sh; select it with FSRL_HIP_LIB=...; its results are invalid)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libfsrl_hip.so"
EXTRA=()
if [ "${1:-}" = "--probes" ]; then
    shift
    OUT="${HERE}/../libfsrl_hip_probe.so"
    EXTRA+=(-DFSRL_PROBES)
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -Wall -Wno-unused-function -Wno-unused-variable \
    "${EXTRA[@]}" "${HERE}/fsrl_hip.hip" -o "${OUT}" "$@"
echo "built ${OUT}"
