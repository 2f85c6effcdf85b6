"""Host-side mirror of the reference's LHP branch with depth propagation.

`LocalHiddenPositiveProjection(cfg)` follows src/modules.py:140-339 for `propagation_strategy == "depth"` (the default):
`forward(code, depth, img=None, attn=None)` returns `projection_head(code)` when `depth` or `attn` is missing (:191-192, the
way the positive image is projected, src/train_segmentation.py:215) and otherwise propagates the code over each position's
nearest points of the depth point cloud before the head (:273-339).  The propagation and its adjoint are the HIP kernels
behind `dg_lhp_forward` / `dg_lhp_backward`; the projection head is two 1x1 convolutions (library GEMMs) owned by torch so
that the caller's optimiser sees its parameters (src/train_segmentation.py:538-543).  The attention strategy ("attn") is not
built.
"""
import torch
import torch.nn as nn

from . import ops


class _DepthPropagation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, code, depth):
        out, points, stats = ops.lhp_forward(code, depth)
        ctx.save_for_backward(points, stats)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        points, stats = ctx.saved_tensors
        return ops.lhp_backward(grad_out.contiguous(), points, stats), None


def propagate_depth(code: torch.Tensor, depth: torch.Tensor) -> torch.Tensor:
    """code_mixed of forward_depth (src/modules.py:279-335): (B,D,h,w) -> (B,D,h,w); differentiable w.r.t. `code`."""
    assert code.shape[0] == depth.shape[0], "Batch size of code and depth must be the same."       # src/modules.py:275
    return _DepthPropagation.apply(code, depth)


class LocalHiddenPositiveProjection(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dim = cfg.dim
        self.propagation_strategy = getattr(cfg, "propagation_strategy", "depth")                   # src/modules.py:144-148
        self.projection_head = nn.Sequential(nn.Conv2d(self.dim, self.dim, (1, 1)), nn.ReLU(),
                                             nn.Conv2d(self.dim, self.dim, (1, 1)))

    def forward(self, code, depth=None, img=None, attn=None):
        if depth is None or attn is None:
            return self.projection_head(code)
        if self.propagation_strategy == "depth":
            return self.forward_depth(code, depth, img)
        if self.propagation_strategy == "attn":
            raise NotImplementedError("depthg_amd: propagation_strategy='attn' is not part of the built path (SURVEY.md 8(f) N3)")
        raise ValueError("Unknown propagation strategy: {}".format(self.propagation_strategy))

    def forward_depth(self, code, depth, img=None):
        return self.projection_head(propagate_depth(code, depth))
