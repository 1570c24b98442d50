"""Host-side data plumbing: a minimal Batch, the proxy of the HIP-resident store, the collector."""
from fsrl_amd._lazy import install

install(__name__, globals(), {
    "Batch": "batch",
    "HipVectorReplayBuffer": "buffer",
    "FastCollector": "fast_collector",
})
