"""The segmentation head and the probes around the correlation loss, on the HIP library.

    ProjectionHead      DinoFeaturizer's trainable part (src/modules.py:75-88, 122-132): `cluster1` (1x1 conv C -> dim) and, for
                        projection_type "nonlinear", `cluster2` (1x1 conv C -> C, ReLU, 1x1 conv C -> dim), each fed by its own
                        Dropout2d(p=.1) draw of the backbone features, plus the third draw the features themselves get when
                        cfg.dropout.  One fused launch (dg_head_forward): the fp32 features are read once; parameters keep the
                        reference's names (`cluster1.0.weight`, `cluster2.2.bias`, ...), so its checkpoints load unchanged.
    ClusterLookup       src/modules.py:647-675 (dg_cluster_lookup_forward / _backward)
    probe_cross_entropy the linear probe's loss, src/train_segmentation.py:427-434: resize of the probe's logits to the label
                        resolution + cross entropy over the labelled pixels in one kernel (dg_probe_ce_forward / _backward)
All arithmetic runs in the library; CPU tensors raise (there is no eager path).
"""
import ctypes
import math

import torch
import torch.nn as nn

from . import _lib
from .ops import _empty, _ptr, _stream


def _gpu32(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"depthg_amd: `{name}` must live on the GPU (got {t.device}); there is no CPU path")
    return t.detach().to(torch.float32).contiguous()


def _check_keeps(keeps, rows, C, device, what):
    """The library receives the keep flags as raw pointers and indexes keep[b * C + k] for b < rows: a mask of another shape
    (the (B, C) per-pass masks handed to the (2B, C) pair call), dtype, device or stride would be read out of bounds or as
    garbage.  Refuse it here.  Returns the 3-tuple (None entries stay None)."""
    if keeps is None:
        return (None, None, None)
    if not isinstance(keeps, (tuple, list)) or len(keeps) != 3:
        raise ValueError(f"depthg_amd: `keeps` of {what} must be a 3-tuple (cluster1's, cluster2's, the returned feats' Dropout2d "
                         f"keep flags; None where that draw does not exist), got {type(keeps).__name__}")
    for i, k in enumerate(keeps):
        if k is None:
            continue
        if not isinstance(k, torch.Tensor):
            raise ValueError(f"depthg_amd: keeps[{i}] of {what} must be a tensor or None, got {type(k).__name__}")
        if tuple(k.shape) != (rows, C):
            raise ValueError(f"depthg_amd: keeps[{i}] of {what} must have shape ({rows}, {C}) - one flag per (image, channel) - "
                             f"got {tuple(k.shape)}")
        if k.dtype != torch.float32:
            raise ValueError(f"depthg_amd: keeps[{i}] of {what} must be float32 0/1 flags, got {k.dtype}")
        if k.device != device:
            raise ValueError(f"depthg_amd: keeps[{i}] of {what} lives on {k.device}, the features on {device}")
        if not k.is_contiguous():
            raise ValueError(f"depthg_amd: keeps[{i}] of {what} must be contiguous (strides {k.stride()})")
    return tuple(keeps)


class _HeadFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, keeps, scale, want_feats, w1, b1, w2a, b2a, w2b, b2b):
        lib = _lib.load()
        f = _gpu32(feat, "image_feat")
        B, C, h, w = f.shape
        D, P, dev = w1.shape[0], h * w, f.device
        nonlinear = w2a is not None
        if ctx.needs_input_grad[0]:
            # (the backbone is frozen in the reference - DinoFeaturizer runs it under no_grad, src/modules.py:96 - and no
            #  d/d image_feat kernel exists: refuse rather than drop that gradient silently)
            raise RuntimeError("depthg_amd: ProjectionHead received features that require grad; the head has no gradient with "
                               "respect to its input (frozen backbone) - detach them")
        # grad mode and requires_grad of the six parameters, as autograd sees them: under torch.no_grad() / eval inference the
        # bf16 hidden tile (19 MB at the headline shape) is neither written nor kept
        need_grad = any(ctx.needs_input_grad[4:])
        code = _empty((B, D, h, w), torch.float32, dev)
        feats_out = _empty((B, C, h, w), torch.float32, dev) if want_feats else None
        hidden = _empty((B, C, P), torch.bfloat16, dev) if (nonlinear and need_grad) else None
        k1, k2, k3 = _check_keeps(keeps, B, C, dev, "ProjectionHead.forward")
        W = lambda t: _gpu32(t, "head parameter").reshape(t.shape[0], -1) if t is not None else None
        w1c, w2ac, w2bc = W(w1), W(w2a), W(w2b)
        wscratch = _empty((lib.dg_head_weights_bytes(C, D),), torch.uint8, dev)     # bf16 copies of the weights (first launch)
        rc = lib.dg_head_forward(B, C, D, P, _ptr(f), _ptr(w1c), _ptr(_gpu32(b1, "bias")), _ptr(w2ac),
                                 _ptr(_gpu32(b2a, "bias")) if nonlinear else None, _ptr(w2bc),
                                 _ptr(_gpu32(b2b, "bias")) if nonlinear else None, _ptr(k1), _ptr(k2), _ptr(k3), float(scale),
                                 _ptr(code), _ptr(feats_out), _ptr(hidden), _ptr(wscratch), _stream(dev))
        _lib.check(rc, "dg_head_forward")
        ctx.dims, ctx.scale, ctx.nonlinear = (B, C, D, P), float(scale), nonlinear
        ctx.shapes = tuple(t.shape if t is not None else None for t in (w1, b1, w2a, b2a, w2b, b2b))
        ctx.save_for_backward(f, k1, k2, hidden, wscratch)
        if feats_out is not None:
            ctx.mark_non_differentiable(feats_out)
        return code, feats_out

    @staticmethod
    def backward(ctx, gcode, _gfeats):
        lib = _lib.load()
        f, k1, k2, hidden, wscratch = ctx.saved_tensors
        B, C, D, P = ctx.dims
        dev = f.device
        if ctx.nonlinear and hidden is None:
            raise RuntimeError("depthg_amd: head backward without the saved hidden activations")
        g = _gpu32(gcode, "grad_code")
        nb = lib.dg_head_workspace_bytes(B, C, D, P)
        ws = _empty(nb, torch.uint8, dev)
        gw1, gb1 = _empty((D, C), torch.float32, dev), _empty((D,), torch.float32, dev)
        gw2a = gb2a = gw2b = gb2b = None
        if ctx.nonlinear:
            gw2a, gb2a = _empty((C, C), torch.float32, dev), _empty((C,), torch.float32, dev)
            gw2b, gb2b = _empty((D, C), torch.float32, dev), _empty((D,), torch.float32, dev)
        rc = lib.dg_head_backward(B, C, D, P, _ptr(f), _ptr(k1), _ptr(k2), ctx.scale, _ptr(hidden), _ptr(wscratch),
                                  _ptr(g), _ptr(gw1), _ptr(gb1), _ptr(gw2a), _ptr(gb2a), _ptr(gw2b), _ptr(gb2b), _ptr(ws), nb,
                                  _stream(dev))
        _lib.check(rc, "dg_head_backward")
        sh = ctx.shapes
        R = lambda t, i: t.reshape(sh[i]) if t is not None else None
        return (None, None, None, None, R(gw1, 0), R(gb1, 1), R(gw2a, 2), R(gb2a, 3), R(gw2b, 4), R(gb2b, 5))


class _HeadPairFunction(torch.autograd.Function):
    """Both featurizer passes of a step (img, img_pos) through ONE set of launches (dg_head_forward_pair / dg_head_backward_pair):
    the same numbers as two _HeadFunction calls whose weight gradients autograd adds up - without the six additions, with half
    the launches, and with no concatenated copy of the features."""

    @staticmethod
    def forward(ctx, feat, feat_pos, keeps, scale, want_feats, w1, b1, w2a, b2a, w2b, b2b):
        lib = _lib.load()
        f, fp = _gpu32(feat, "image_feat"), _gpu32(feat_pos, "image_feat_pos")
        if f.shape != fp.shape or f.device != fp.device:
            raise ValueError(f"depthg_amd: the two passes' features must match: {tuple(f.shape)} on {f.device} and {tuple(fp.shape)} on {fp.device}")
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            raise RuntimeError("depthg_amd: ProjectionHead received features that require grad; the head has no gradient with "
                               "respect to its input (frozen backbone) - detach them")
        B, C, h, w = f.shape
        D, P, dev = w1.shape[0], h * w, f.device
        nonlinear = w2a is not None
        need_grad = any(ctx.needs_input_grad[5:])
        code, code_pos = _empty((B, D, h, w), torch.float32, dev), _empty((B, D, h, w), torch.float32, dev)
        fo = _empty((B, C, h, w), torch.float32, dev) if want_feats else None
        fo_pos = _empty((B, C, h, w), torch.float32, dev) if want_feats else None
        hidden = _empty((2 * B, C, P), torch.bfloat16, dev) if (nonlinear and need_grad) else None
        k1, k2, k3 = _check_keeps(keeps, 2 * B, C, dev, "ProjectionHead.forward_pair")   # (2B, C) each: the first pass's rows first
        W = lambda t: _gpu32(t, "head parameter").reshape(t.shape[0], -1) if t is not None else None
        w1c, w2ac, w2bc = W(w1), W(w2a), W(w2b)
        wscratch = _empty((lib.dg_head_weights_bytes(C, D),), torch.uint8, dev)
        rc = lib.dg_head_forward_pair(B, C, D, P, _ptr(f), _ptr(fp), _ptr(w1c), _ptr(_gpu32(b1, "bias")), _ptr(w2ac),
                                      _ptr(_gpu32(b2a, "bias")) if nonlinear else None, _ptr(w2bc),
                                      _ptr(_gpu32(b2b, "bias")) if nonlinear else None, _ptr(k1), _ptr(k2), _ptr(k3), float(scale),
                                      _ptr(code), _ptr(code_pos), _ptr(fo), _ptr(fo_pos), _ptr(hidden), _ptr(wscratch), _stream(dev))
        _lib.check(rc, "dg_head_forward_pair")
        ctx.dims, ctx.scale, ctx.nonlinear = (B, C, D, P), float(scale), nonlinear
        ctx.shapes = tuple(t.shape if t is not None else None for t in (w1, b1, w2a, b2a, w2b, b2b))
        ctx.save_for_backward(f, fp, k1, k2, hidden, wscratch)
        ctx.set_materialize_grads(False)
        if fo is not None:
            ctx.mark_non_differentiable(fo, fo_pos)
        return code, code_pos, fo, fo_pos

    @staticmethod
    def backward(ctx, gcode, gcode_pos, _gf, _gfp):
        lib = _lib.load()
        f, fp, k1, k2, hidden, wscratch = ctx.saved_tensors
        B, C, D, P = ctx.dims
        dev = f.device
        if ctx.nonlinear and hidden is None:
            raise RuntimeError("depthg_amd: head backward without the saved hidden activations")
        # (a pass whose code took no part in the loss contributes nothing: a zero upstream)
        g = _gpu32(gcode, "grad_code") if gcode is not None else torch.zeros((B, D, P), dtype=torch.float32, device=dev)
        gp = _gpu32(gcode_pos, "grad_code_pos") if gcode_pos is not None else torch.zeros((B, D, P), dtype=torch.float32, device=dev)
        nb = lib.dg_head_workspace_bytes(2 * B, C, D, P)
        ws = _empty(nb, torch.uint8, dev)
        gw1, gb1 = _empty((D, C), torch.float32, dev), _empty((D,), torch.float32, dev)
        gw2a = gb2a = gw2b = gb2b = None
        if ctx.nonlinear:
            gw2a, gb2a = _empty((C, C), torch.float32, dev), _empty((C,), torch.float32, dev)
            gw2b, gb2b = _empty((D, C), torch.float32, dev), _empty((D,), torch.float32, dev)
        rc = lib.dg_head_backward_pair(B, C, D, P, _ptr(f), _ptr(fp), _ptr(k1), _ptr(k2), ctx.scale, _ptr(hidden), _ptr(wscratch),
                                       _ptr(g), _ptr(gp), _ptr(gw1), _ptr(gb1), _ptr(gw2a), _ptr(gb2a), _ptr(gw2b), _ptr(gb2b),
                                       _ptr(ws), nb, _stream(dev))
        _lib.check(rc, "dg_head_backward_pair")
        sh = ctx.shapes
        R = lambda t, i: t.reshape(sh[i]) if t is not None else None
        return (None, None, None, None, None, R(gw1, 0), R(gb1, 1), R(gw2a, 2), R(gb2a, 3), R(gw2b, 4), R(gb2b, 5))


def draw_keep_masks_pair(B, C, device, p=0.1, use=(True, True, True)):
    """The six Dropout2d draws of a step's two featurizer passes in the reference's order - those of pass 1 (cluster1's input,
    cluster2's input, the returned feats), then those of pass 2 - as THREE (2B, C) tensors whose halves are drawn in place: the
    torch generator advances exactly as it does for two draw_keep_masks calls, and nothing is concatenated."""
    ks = [torch.empty(2 * B, C, device=device, dtype=torch.float32) if u else None for u in use]
    for half in (slice(0, B), slice(B, 2 * B)):
        for k in ks:
            if k is not None:
                k[half].bernoulli_(1.0 - p)
    return tuple(ks)


def draw_keep_masks(B, C, device, p=0.1, use=(True, True, True)):
    """The Dropout2d draws of one featurizer pass, in the reference's order (cluster1's input, cluster2's input, the returned
    feats; src/modules.py:122-132): (B, C) keep flags, one bernoulli_(1 - p) each, as F.dropout2d draws its (B, C, 1, 1) noise.
    Only the uses that exist draw (`use`: cluster1, cluster2 = projection_type "nonlinear", feats = cfg.dropout) - the reference
    calls Dropout2d exactly there, so the torch generator advances as it does there; the others are None."""
    return tuple(torch.empty(B, C, device=device, dtype=torch.float32).bernoulli_(1.0 - p) if u else None for u in use)


def run_head(cluster1, cluster2, image_feat, training, feats_dropout, p=0.1, keeps=None):
    """(code, feats) of one featurizer pass (src/modules.py:122-137) from the reference's modules: `cluster1` = Sequential(Conv2d),
    `cluster2` = Sequential(Conv2d, ReLU, Conv2d) or None (projection_type "linear").  Training: three Dropout2d draws (`keeps`
    or drawn here), feats = Dropout2d(image_feat) when `feats_dropout` (cfg.dropout); eval: no dropout, feats = image_feat."""
    B, C = image_feat.shape[:2]
    nl = cluster2 is not None
    if training:
        if keeps is None:
            keeps = draw_keep_masks(B, C, image_feat.device, p, use=(True, nl, bool(feats_dropout)))
        k1, k2, k3 = keeps
        keeps = (k1, k2 if nl else None, k3 if feats_dropout else None)
    else:
        keeps = None                                                                                  # eval: Dropout2d is the identity
    c1 = cluster1[0]
    c2a, c2b = (cluster2[0], cluster2[2]) if nl else (None, None)
    want_feats = bool(training and feats_dropout)
    code, feats = _HeadFunction.apply(image_feat, keeps, 1.0 / (1.0 - p), want_feats, c1.weight, c1.bias,
                                      c2a.weight if nl else None, c2a.bias if nl else None,
                                      c2b.weight if nl else None, c2b.bias if nl else None)
    return code, (feats if want_feats else image_feat)


def run_head_pair(cluster1, cluster2, image_feat, image_feat_pos, training, feats_dropout, p=0.1, keeps=None, defer_feats_dropout=False):
    """run_head for the two passes of a training step at once: ((code, feats), (code_pos, feats_pos)).  `keeps`: three (2B, C)
    tensors (draw_keep_masks_pair) or None (drawn here, in the order two run_head calls would draw).  `defer_feats_dropout`: the
    Dropout2d of the returned feats is drawn as always but NOT applied - `feats` / `feats_pos` come back as ops.DeferredDropout
    (the input maps + their keep flags), which ContrastiveCorrelationLoss applies inside its operand preparation: the dropped
    tensors (38.5 MB each at the headline shape) are neither written here nor read there."""
    B, C = image_feat.shape[:2]
    nl = cluster2 is not None
    if training:
        if keeps is None:
            keeps = draw_keep_masks_pair(B, C, image_feat.device, p, use=(True, nl, bool(feats_dropout)))
        k1, k2, k3 = keeps
        keeps = (k1, k2 if nl else None, k3 if feats_dropout else None)
    else:
        keeps = None
    c1 = cluster1[0]
    c2a, c2b = (cluster2[0], cluster2[2]) if nl else (None, None)
    want_feats = bool(training and feats_dropout)
    deferred = None
    if want_feats and defer_feats_dropout:
        deferred, want_feats = keeps[2], False
        keeps = (keeps[0], keeps[1], None)
    code, code_pos, feats, feats_pos = _HeadPairFunction.apply(
        image_feat, image_feat_pos, keeps, 1.0 / (1.0 - p), want_feats, c1.weight, c1.bias,
        c2a.weight if nl else None, c2a.bias if nl else None, c2b.weight if nl else None, c2b.bias if nl else None)
    if deferred is not None:
        from .ops import DeferredDropout
        scale = 1.0 / (1.0 - p)
        return ((code, DeferredDropout(image_feat, deferred[:B], scale)), (code_pos, DeferredDropout(image_feat_pos, deferred[B:], scale)))
    return (code, feats if want_feats else image_feat), (code_pos, feats_pos if want_feats else image_feat_pos)


class ProjectionHead(nn.Module):
    """cluster1 / cluster2 of DinoFeaturizer with the reference's module and parameter names; forward(image_feat, feats_dropout,
    keeps) -> (code, feats): feats = Dropout2d(image_feat) when `feats_dropout` (cfg.dropout) and the module trains, else
    image_feat itself."""

    def __init__(self, n_feats: int, dim: int, projection_type="nonlinear", p: float = 0.1):
        super().__init__()
        self.n_feats, self.dim, self.proj_type, self.p = n_feats, dim, projection_type, p
        self.cluster1 = nn.Sequential(nn.Conv2d(n_feats, dim, (1, 1)))                                   # make_clusterer, :75-77
        if projection_type == "nonlinear":                                                                # make_nonlinear_clusterer, :79-83
            self.cluster2 = nn.Sequential(nn.Conv2d(n_feats, n_feats, (1, 1)), nn.ReLU(), nn.Conv2d(n_feats, dim, (1, 1)))

    def forward(self, image_feat, feats_dropout=True, keeps=None):
        if self.proj_type is None:                                                                        # :125-126: code = image_feat
            feats = image_feat
            if self.training and feats_dropout:                                                           # :127-129: feats = Dropout2d(image_feat)
                if keeps is not None:
                    _check_keeps(keeps, image_feat.shape[0], image_feat.shape[1], image_feat.device, "ProjectionHead.forward")
                k3 = keeps[2] if keeps is not None and keeps[2] is not None else draw_keep_masks(image_feat.shape[0], image_feat.shape[1], image_feat.device,
                                                                         self.p, use=(False, False, True))[2]
                feats = image_feat * (k3 * (1.0 / (1.0 - self.p)))[:, :, None, None]
            return image_feat, feats
        return run_head(self.cluster1, getattr(self, "cluster2", None) if self.proj_type == "nonlinear" else None, image_feat,
                        self.training, feats_dropout, self.p, keeps)

    def forward_pair(self, image_feat, image_feat_pos, feats_dropout=True, keeps=None, defer_feats_dropout=False):
        """forward(image_feat) and forward(image_feat_pos) of one training step (src/train_segmentation.py:303-306) as one set of
        launches: ((code, feats), (code_pos, feats_pos)), the same values and - after backward - the same parameter gradients.
        `defer_feats_dropout`: see run_head_pair (feats / feats_pos as ops.DeferredDropout for the loss to apply)."""
        if self.proj_type is None:
            ka = kb = None
            if keeps is not None:
                B = image_feat.shape[0]
                _check_keeps(keeps, 2 * B, image_feat.shape[1], image_feat.device, "ProjectionHead.forward_pair")
                if keeps[2] is not None:
                    ka, kb = (None, None, keeps[2][:B]), (None, None, keeps[2][B:])
            return self.forward(image_feat, feats_dropout, ka), self.forward(image_feat_pos, feats_dropout, kb)
        return run_head_pair(self.cluster1, getattr(self, "cluster2", None) if self.proj_type == "nonlinear" else None, image_feat,
                             image_feat_pos, self.training, feats_dropout, self.p, keeps, defer_feats_dropout)


class _ClusterFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, clusters, alpha, want_logp):
        lib = _lib.load()
        xx, cc = _gpu32(x, "x"), _gpu32(clusters, "clusters")
        B, D, h, w = xx.shape
        n, P, dev = cc.shape[0], h * w, xx.device
        inner = _empty((B, n, h, w), torch.float32, dev)
        probs = _empty((B, n, h, w), torch.float32, dev)
        logp = _empty((B, n, h, w), torch.float32, dev) if want_logp else None
        loss = _empty((1,), torch.float32, dev)
        scratch = _empty((B * ((P + 255) // 256),), torch.float32, dev)
        a = float("nan") if alpha is None else float(alpha)
        rc = lib.dg_cluster_lookup_forward(_ptr(xx), _ptr(cc), a, B, D, n, P, _ptr(inner), _ptr(probs), _ptr(logp), _ptr(loss),
                                           _ptr(scratch), _stream(dev))
        _lib.check(rc, "dg_cluster_lookup_forward")
        ctx.alpha, ctx.dims, ctx.x_grad = a, (B, D, n, P), x.requires_grad
        ctx.save_for_backward(xx, cc, inner)
        ctx.mark_non_differentiable(probs)
        if logp is not None:
            return loss[0], probs, logp
        return loss[0], probs, None

    @staticmethod
    def backward(ctx, gloss, _gprobs, glogp):
        if glogp is not None:
            raise RuntimeError("depthg_amd: ClusterLookup(log_probs=True) is an evaluation output and carries no gradient")
        lib = _lib.load()
        xx, cc, inner = ctx.saved_tensors
        B, D, n, P = ctx.dims
        dev = xx.device
        g = _gpu32(gloss, "grad_loss").reshape(1)
        gc = _empty((n, D), torch.float32, dev)
        gx = _empty(tuple(xx.shape), torch.float32, dev) if ctx.x_grad else None
        scratch = _empty((B * n * P + B * ((P + 63) // 64) * n * D,), torch.float32, dev)
        rc = lib.dg_cluster_lookup_backward(_ptr(xx), _ptr(cc), _ptr(inner), ctx.alpha, _ptr(g), B, D, n, P, _ptr(gc), _ptr(gx),
                                            _ptr(scratch), _stream(dev))
        _lib.check(rc, "dg_cluster_lookup_backward")
        return gx, gc, None, None


class ClusterLookup(nn.Module):
    """Cosine cluster probe (src/modules.py:647-675): `clusters` (n_classes, dim) centres; forward(x, alpha, log_probs=False)
    returns (cluster_loss, cluster_probs) - hard one-hot assignment when alpha is None, softmax(alpha * similarity) otherwise -
    or the log-probabilities alone."""

    def __init__(self, dim: int, n_classes: int):
        super().__init__()
        self.n_classes, self.dim = n_classes, dim
        self.clusters = nn.Parameter(torch.randn(n_classes, dim))

    def reset_parameters(self):
        with torch.no_grad():
            self.clusters.copy_(torch.randn(self.n_classes, self.dim))

    def forward(self, x, alpha, log_probs=False):
        if log_probs:
            if alpha is None:
                raise TypeError("log_probs=True needs alpha (the reference multiplies by it)")
            return _ClusterFunction.apply(x, self.clusters, alpha, True)[2]
        loss, probs, _ = _ClusterFunction.apply(x, self.clusters, alpha, False)
        return loss, probs


class _ProbeCeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, label):
        lib = _lib.load()
        lg = _gpu32(logits, "linear_logits")
        if not label.is_cuda:
            raise RuntimeError("depthg_amd: `label` must live on the GPU; there is no CPU path")
        lab = label.detach().to(torch.int64).contiguous()
        B, n, h, w = lg.shape
        H, W = lab.shape[-2:]
        lab = lab.reshape(B, H, W)
        dev = lg.device
        out3 = _empty((3,), torch.float32, dev)
        scratch = _empty((2 * B * H,), torch.float32, dev)
        rc = lib.dg_probe_ce_forward(_ptr(lg), _ptr(lab), B, n, h, w, H, W, _ptr(out3), _ptr(scratch), _stream(dev))
        _lib.check(rc, "dg_probe_ce_forward")
        ctx.dims = (B, n, h, w, H, W)
        ctx.save_for_backward(lg, lab, out3)
        return out3[2]

    @staticmethod
    def backward(ctx, gloss):
        lib = _lib.load()
        lg, lab, out3 = ctx.saved_tensors
        B, n, h, w, H, W = ctx.dims
        g = _gpu32(gloss, "grad_loss").reshape(1)
        gl = _empty((B, n, h, w), torch.float32, lg.device)
        rc = lib.dg_probe_ce_backward(_ptr(lg), _ptr(lab), _ptr(out3), _ptr(g), B, n, h, w, H, W, _ptr(gl), _stream(lg.device))
        _lib.check(rc, "dg_probe_ce_backward")
        return gl, None


def probe_cross_entropy(linear_logits, label):
    """`CrossEntropyLoss()(interpolate(linear_logits, label.shape[-2:], 'bilinear', align_corners=False)[mask], label[mask])`
    with mask = (label >= 0) & (label < n_classes)  (src/train_segmentation.py:421-434), without the label-resolution logits."""
    return _ProbeCeFunction.apply(linear_logits, label)
