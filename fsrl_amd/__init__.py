"""fsrl_amd -- MI355X-native policy-update engine behind FSRL's Python seam.

The compute path is `libfsrl_hip.so` (hand-written HIP for gfx950, built in-tree by
`fsrl_amd/csrc/build.sh` / `__graft_entry__.build()`), reached through the C ABI declared
in `include/fsrl_hip.h`.  There is NO CPU fallback: importing `fsrl_amd.engine` without the
library raises, and constructing an engine without a GPU raises.
"""
__version__ = "0.1.0"
