"""Epoch loops that call the hot path (fsrl.trainer's three names)."""
from fsrl_amd._lazy import install

install(__name__, globals(), {
    "BaseTrainer": "base_trainer",
    "OnpolicyTrainer": "onpolicy",
    "OffpolicyTrainer": "offpolicy",
})
