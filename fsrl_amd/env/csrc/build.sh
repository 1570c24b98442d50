#!/bin/bash
# Build libfsrl_env.so (plain C, gcc): the futex handshake of the worker-process vector env.  In-tree output so it
# travels to the GPU box with the repo snapshot.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../../libfsrl_env.so"
gcc -O2 -std=c11 -fPIC -shared -Wall -Wextra "${HERE}/fsrl_env.c" -o "${OUT}"
echo "built ${OUT}"
