"""depthg_amd - MI355X (gfx950) implementation of DepthG's feature-correlation loss hot path.

Public surface (mirrors the reference's Python operator surface for this path):
    ContrastiveCorrelationLoss   drop-in for src/modules.py:1221-1367
    depth_decay                  scalar decay schedules (src/depth_decay_modules.py) + the live legacy decay
    ops                          thin ctypes binding of the C ABI in include/depthg_corr.h
"""
from .loss import ContrastiveCorrelationLoss  # noqa: F401
from . import depth_decay  # noqa: F401

__all__ = ["ContrastiveCorrelationLoss", "depth_decay"]
