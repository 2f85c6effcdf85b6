"""Host-side mirror of the reference's LHP branch.

`LocalHiddenPositiveProjection(cfg)` follows src/modules.py:140-339: `forward(code, depth, img=None, attn=None)` returns
`projection_head(code)` when `depth` or `attn` is missing (:191-192, the way the positive image is projected,
src/train_segmentation.py:215) and otherwise propagates the code before the head - over each position's nearest points of the
depth point cloud (`propagation_strategy == "depth"`, the default, :273-339) or over the backbone's last self-attention
(`"attn"`, :235-271).  `OriginalLocalHiddenPositiveProjection(cfg)` follows :342-487 (the variant `train_segmentation.py:83-85`
picks for experiment names containing "lhp_original"): the same two sources restricted to the clipped 3x3 neighbourhood.
The propagations and their adjoints are HIP kernels (`dg_lhp_forward/backward`, `dg_lhp_map_forward/backward`); the projection
head is two 1x1 convolutions (library GEMMs) owned by torch.  As in the reference, no optimiser steps them:
`configure_optimizers` hands `net_optim` the parameters of `self.net` only (src/train_segmentation.py:537-547), so the LHP
projection head keeps its initial weights; gradients do reach it and flow through it into the code.

Reference behaviour kept on purpose:
  * `divide_num` of both classes is all zero: the constructor re-creates it inside its loop and never fills it (:160,187 /
    :354,382), so the Original variants return `sum / 0` (inf / nan).  It is a buffer here, zero by default;
    `neighbour_counts(sz)` is the table the code it was taken from intended, for callers that repair it.
  * `forward_attn_lhp` (:200-233) has no call site in the reference and is not built.
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


class _MapPropagation(torch.autograd.Function):
    """dg_lhp_map_forward / dg_lhp_map_backward; no gradient for the attention / depth / divisors (frozen backbone, data)."""

    @staticmethod
    def forward(ctx, code, source, divide, mode):
        if mode == ops.LHP_ORIG_DEPTH:
            out, wmap = ops.lhp_map_forward(mode, code, depth=source, divide=divide)
        else:
            out, wmap = ops.lhp_map_forward(mode, code, attn=source, divide=divide)
        ctx.mode = mode
        ctx.has_divide = divide is not None
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(*((wmap, divide) if ctx.has_divide else (wmap,)))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        saved = ctx.saved_tensors
        divide = saved[1] if ctx.has_divide else None
        return ops.lhp_map_backward(ctx.mode, grad_out.contiguous(), saved[0], divide), None, None, None


def propagate_depth(code: torch.Tensor, depth: torch.Tensor) -> torch.Tensor:
    """code_mixed of forward_depth (src/modules.py:279-335): (B,D,h,w) -> (B,D,h,w); differentiable w.r.t. `code`."""
    assert code.shape[0] == depth.shape[0], "Batch size of code and depth must be the same."       # src/modules.py:275
    return _DepthPropagation.apply(code, depth)


def propagate_attn(code: torch.Tensor, attn: torch.Tensor) -> torch.Tensor:
    """code_mixed of forward_attn (src/modules.py:235-269); attn = the backbone's last self-attention (B,heads,P+1,P+1)."""
    assert code.shape[0] == attn.shape[0], "Batch size of code and depth must be the same."        # src/modules.py:236 (sic)
    return _MapPropagation.apply(code, attn, None, ops.LHP_ATTN)


def neighbour_counts(sz: int) -> torch.Tensor:
    """Sizes of the clipped 3x3 neighbourhoods of a sz x sz map, (sz*sz, 1) int64: what `divide_num` holds in the code the
    reference's constructor was taken from (the reference itself leaves zeros)."""
    n = torch.full((sz,), 3, dtype=torch.long)
    n[0] = n[-1] = 2
    return (n[:, None] * n[None, :]).reshape(-1, 1)


def _projection_head(dim):
    return nn.Sequential(nn.Conv2d(dim, dim, (1, 1)), nn.ReLU(), nn.Conv2d(dim, dim, (1, 1)))


class LocalHiddenPositiveProjection(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dim = cfg.dim
        self.propagation_strategy = getattr(cfg, "propagation_strategy", "depth")                   # src/modules.py:144-148
        self.projection_head = _projection_head(self.dim)

    def forward(self, code, depth=None, img=None, attn=None):
        if depth is None or attn is None:
            return self.projection_head(code)
        if self.propagation_strategy == "depth":
            return self.forward_depth(code, depth, img)
        if self.propagation_strategy == "attn":
            return self.forward_attn(code, attn)
        raise ValueError("Unknown propagation strategy: {}".format(self.propagation_strategy))

    def forward_depth(self, code, depth, img=None):
        return self.projection_head(propagate_depth(code, depth))

    def forward_attn(self, code, attn=None):
        return self.projection_head(propagate_attn(code, attn))


class OriginalLocalHiddenPositiveProjection(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dim = cfg.dim
        self.propagation_strategy = cfg.propagation_strategy                                        # src/modules.py:346 (required)
        self.projection_head = _projection_head(self.dim)
        sz = cfg.res // cfg.dino_patch_size                                                         # src/modules.py:355
        self.sz = sz
        self.register_buffer("divide_num", torch.zeros((sz * sz, 1), dtype=torch.long), persistent=False)

    def _check(self, code, other):
        assert code.shape[0] == other.shape[0], "Batch size of code and depth must be the same."
        if code.shape[-2] != self.sz or code.shape[-1] != self.sz:
            # (the reference's (P,P) index mask only broadcasts against maps of res // dino_patch_size positions a side)
            raise RuntimeError(f"code map {tuple(code.shape[-2:])} does not match res // dino_patch_size = {self.sz}")

    def forward(self, code, depth=None, img=None, attn=None):
        if depth is None or attn is None:
            return self.projection_head(code)
        if self.propagation_strategy == "depth":
            return self.forward_depth(code, depth, img)
        if self.propagation_strategy == "attn":
            return self.forward_attn(code, attn)
        raise ValueError("Unknown propagation strategy: {}".format(self.propagation_strategy))

    def forward_attn(self, code, attn=None):
        self._check(code, attn)
        divide = self.divide_num.to(device=code.device, dtype=torch.float32).reshape(-1)
        return self.projection_head(_MapPropagation.apply(code, attn, divide, ops.LHP_ORIG_ATTN))

    def forward_depth(self, code, depth, img=None):
        self._check(code, depth)
        divide = self.divide_num.to(device=code.device, dtype=torch.float32).reshape(-1)
        return self.projection_head(_MapPropagation.apply(code, depth, divide, ops.LHP_ORIG_DEPTH))
