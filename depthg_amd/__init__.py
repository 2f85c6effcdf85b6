"""depthg_amd - MI355X (gfx950) implementation of DepthG's feature-correlation loss hot path.

Public surface (mirrors the reference's Python operator surface for this path):
    ContrastiveCorrelationLoss   drop-in for src/modules.py:1221-1367
    depth_decay                  scalar decay schedules (src/depth_decay_modules.py) + the live legacy decay
    training                     the caller arithmetic around the loss (means, weighted total, log keys;
                                 src/train_segmentation.py:240-350)
    metrics                      UnsupervisedMetrics (src/utils.py:202-319): confusion matrix on the GPU, summed over DP ranks
    knn                          image-level nearest-neighbour table: search, file format, online pick (src/precompute_knns.py)
    lhp                          LocalHiddenPositiveProjection with depth propagation (src/modules.py:140-339)
    segmenter                    the producers and the caller: projection head, stand-in featurizer, cluster probe, one optimisation
                                 step of LitUnsupervisedSegmenter (src/modules.py:19-137,647-675; src/train_segmentation.py:71-462)
    ops                          thin ctypes binding of the C ABI in include/depthg_corr.h
"""
from .loss import ContrastiveCorrelationLoss  # noqa: F401
from . import depth_decay  # noqa: F401
from . import training  # noqa: F401
from . import metrics  # noqa: F401
from . import knn  # noqa: F401
from . import lhp  # noqa: F401
from . import segmenter  # noqa: F401

__all__ = ["ContrastiveCorrelationLoss", "depth_decay", "training", "metrics", "knn", "lhp", "segmenter"]
