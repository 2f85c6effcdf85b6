"""Host mirror of the caller arithmetic around the correlation loss:
`LitUnsupervisedSegmenter.training_step`, reference src/train_segmentation.py:240-350 - the means of the returned
tuple, the log keys, and the weighted correspondence total including the `correspondence_weight - balance` factor
and the optional second (LHP) evaluation of the loss.  Pure host/torch-scalar logic on the tuple the module returns;
nothing here touches the HIP library, so it runs on any device the tuple lives on.

    out = loss_fn(feats, feats_pos, sal, sal_pos, code, code_pos, depth, depth_pos)
    total, logs = correspondence_total(cfg, out)                 # what `loss += ...` adds at :335-350
    total.backward()

`fused_correspondence_total(cfg, loss_fn)` is the same number as formed inside the library (`loss_fn.total`, no torch op at
all; what bench.py times).
"""
from typing import Dict, Optional, Sequence, Tuple

import torch

LOG_KEYS_LOSS = ("loss/pos_intra", "loss/pos_inter", "loss/neg_inter", "loss/depth_feat")
LOG_KEYS_CD = ("cd/pos_intra", "cd/pos_inter", "cd/neg_inter", "cd/depth_feat")


def _balance(cfg) -> float:
    # src/train_segmentation.py:325-328: only the depth branch subtracts the LHP weight
    return float(cfg.lhp_weight) if getattr(cfg, "lhp", False) and getattr(cfg, "lhp_weight_balance", False) else 0.0


def correspondence_total(cfg, out: Sequence[torch.Tensor], lhp_out: Optional[Sequence[torch.Tensor]] = None
                         ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """`out`: the 6- or 8-tuple of ContrastiveCorrelationLoss (8 iff cfg.depth_feat_correlation_loss);
    `lhp_out`: the tuple of the second call on the LHP-projected code when cfg.lhp (src/train_segmentation.py:255-266,
    281-292), else None.  Returns (term added to the training loss, {log key: detached scalar})."""
    depth = len(out) == 8
    if depth != bool(getattr(cfg, "depth_feat_correlation_loss", False)):
        raise ValueError(f"tuple of {len(out)} does not match cfg.depth_feat_correlation_loss")
    pos_intra, pos_inter, neg_inter = out[0].mean(), out[2].mean(), out[4].mean()        # :303-305
    logs = {"loss/pos_intra": pos_intra.detach(), "loss/pos_inter": pos_inter.detach(), "loss/neg_inter": neg_inter.detach(),
            "cd/pos_intra": out[1].mean().detach(), "cd/pos_inter": out[3].mean().detach(), "cd/neg_inter": out[5].mean().detach()}
    core = cfg.pos_inter_weight * pos_inter + cfg.pos_intra_weight * pos_intra + cfg.neg_inter_weight * neg_inter
    if depth:
        depth_feat = out[6].mean()                                                        # :312-316
        logs["loss/depth_feat"] = depth_feat.detach()
        logs["cd/depth_feat"] = out[7].mean().detach()
        total = (core + cfg.depth_feat_weight * depth_feat) * (cfg.correspondence_weight - _balance(cfg))   # :330-333
        if "lhp_original" in str(getattr(cfg, "experiment_name", "")):                    # :335-337: only the LHP terms train,
            total = torch.zeros((), dtype=total.dtype, device=total.device)              # `loss = 0.0` (:336): a nan/inf main term
                                                                                          # is dropped, not turned into 0 * inf
            cfg.lhp_weight = 1.0
    else:
        total = core * cfg.correspondence_weight                                          # :347-349
    if getattr(cfg, "lhp", False) and depth:                                              # :339-343 (depth branch only)
        if lhp_out is None:
            raise ValueError("cfg.lhp is set: pass the tuple of the second loss call on the projected code")
        lhp_depth = lhp_out[6].mean() if len(lhp_out) == 8 else 0.0
        total = total + (cfg.pos_inter_weight * lhp_out[2].mean() + cfg.pos_intra_weight * lhp_out[0].mean() +
                         cfg.neg_inter_weight * lhp_out[4].mean() +
                         cfg.depth_feat_weight * cfg.lhp_depth_weight * lhp_depth) * cfg.lhp_weight
    return total, logs


def correspondence_weights(cfg, depth: bool, device, full: bool = False) -> torch.Tensor:
    """fp32 [4] weights of (intra, inter, neg, depth) loss means in the total, in the order of the fused output vector;
    `full`: zero-padded to the length of the fused output vector, so that `dot(loss_fn.scalars, w)` needs no slice (and its backward
    no zero-fill + copy)."""
    scale = (cfg.correspondence_weight - _balance(cfg)) if depth else cfg.correspondence_weight
    w = torch.tensor([cfg.pos_intra_weight, cfg.pos_inter_weight, cfg.neg_inter_weight,
                      cfg.depth_feat_weight if depth else 0.0] + ([0.0] * 5 if full else []), dtype=torch.float32)
    return (w * scale).to(device)


def fused_correspondence_total(cfg, loss_fn) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """Same value as correspondence_total(cfg, out) for the last call of `loss_fn` (without LHP): the library's own
    weighted total."""
    s = loss_fn.scalars
    depth = bool(getattr(cfg, "depth_feat_correlation_loss", False))
    total = loss_fn.total                      # formed by the library with the weights of cfg (DG_OUT_TOTAL)
    d = s.detach()
    logs = {k: d[i] for i, k in enumerate(LOG_KEYS_LOSS[:4 if depth else 3])}
    logs.update({k: d[4 + i] for i, k in enumerate(LOG_KEYS_CD[:4 if depth else 3])})
    return total, logs
