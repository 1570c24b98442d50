"""Host-side data plumbing: minimal Batch, the HIP-resident buffer proxy and the collector."""
from fsrl_amd.data.batch import Batch
from fsrl_amd.data.buffer import HipVectorReplayBuffer
from fsrl_amd.data.fast_collector import FastCollector

__all__ = ["Batch", "HipVectorReplayBuffer", "FastCollector"]
