"""vae_amd -- MI355X-native Variational Factorization Machine training step.

The hot path of jilljenn/vae's `vfm-torch.py` (gathers -> reparameterised sample -> FM
interaction -> ELBO, forward and backward) as hand-written HIP kernels for gfx950 behind a C ABI
(`include/vfm_hip.h`, `libvfm_hip.so`), with the reference's fit()/predict() behaviour on top.
"""
from . import _lib  # noqa: F401
from .ops import Spec, BatchPlan, ElboFunction  # noqa: F401

__all__ = ["Spec", "BatchPlan", "ElboFunction"]
