#!/bin/bash
# Build libfsrl_hip.so for gfx950 (cross-compiles without a GPU).  In-tree output so the
# .so travels to the GPU box with the repo snapshot.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libfsrl_hip.so"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -Wall -Wno-unused-function -Wno-unused-variable \
    "${HERE}/fsrl_hip.hip" -o "${OUT}" "$@"
echo "built ${OUT}"
