"""Producers and caller of the correlation loss: the segmentation head, the probes and one optimisation step
(SURVEY.md section 8 rows A14, N1 and the step-level caller around A13).

Mirrors, with the reference's names, parameter layout and arithmetic:
    StandInFeaturizer       a featurizer with DinoFeaturizer's output contract (src/modules.py:90-137: train -> (feats, code,
                            attn), eval -> (feats, code); feats get a third Dropout2d mask when cfg.dropout).  The frozen DINO
                            ViT itself is out of scope (north_star: "the Python host calls PyTorch-ROCm for the ViT forward") and
                            its weights cannot be fetched here, so the backbone is a frozen random patch embedding of the same
                            geometry (patch size, n_feats, no gradient) - any module returning (B, n_feats, H/p, W/p) can be
                            passed instead.  Its head (`cluster1` / `cluster2`, the three Dropout2d draws) is ONE fused HIP launch
                            (depthg_amd/head.py run_head -> dg_head_forward / dg_head_backward).
    UnsupervisedSegmenter   LitUnsupervisedSegmenter without Lightning (src/train_segmentation.py:71-158, 169-462): attributes
                            net / train_cluster_probe / cluster_probe / linear_probe / contrastive_corr_loss_fn / cfg /
                            n_classes / use_depth, forward(x) = net(x)[1], configure_optimizers() -> three Adams,
                            training_step(batch, batch_idx) with manual optimisation: two featurizer passes, the HIP correlation
                            loss (and the second LHP call when cfg.lhp), the weighted total, the live legacy decay of the cfg
                            scalars, linear-probe cross entropy and cluster-probe loss on the detached code, backward, three steps.
ProjectionHead / ClusterLookup (src/modules.py:647-675) / probe_cross_entropy live in depthg_amd/head.py and are re-exported
here.  The head, the probes' losses and everything the correlation loss does run in the HIP library; what stays torch is the
frozen backbone, the 27 x dim linear-probe convolution and the three Adams.  Under data parallelism `all_reduced_parameters()` is
what `parallel.GradBucket` all-reduces.
"""
from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn

from .depth_decay import legacy_decay_step
from .head import ClusterLookup, ProjectionHead, probe_cross_entropy, run_head, run_head_pair
from .lhp import LocalHiddenPositiveProjection, OriginalLocalHiddenPositiveProjection
from .loss import ContrastiveCorrelationLoss
from .training import correspondence_total


class StandInFeaturizer(nn.Module):
    """DinoFeaturizer's contract with a frozen stand-in backbone (see the module docstring)."""

    def __init__(self, dim: int, cfg, backbone: Optional[nn.Module] = None):
        super().__init__()
        self.cfg, self.dim = cfg, dim
        self.patch_size = int(cfg.dino_patch_size)
        arch = str(getattr(cfg, "model_type", "vit_small"))
        self.n_feats = 384 if "small" in arch else 768          # src/modules.py:70-73
        if backbone is None:
            backbone = nn.Conv2d(3, self.n_feats, self.patch_size, stride=self.patch_size)
        self.model = backbone
        for p in self.model.parameters():                       # frozen, as the DINO ViT (:34-35)
            p.requires_grad = False
        self.dropout = nn.Dropout2d(p=.1)
        head = ProjectionHead(self.n_feats, dim, getattr(cfg, "projection_type", "nonlinear"))   # (modules only: run_head does the work)
        self.cluster1 = head.cluster1                           # registered under the reference's names
        if hasattr(head, "cluster2"):
            self.cluster2 = head.cluster2
        self.proj_type = head.proj_type

    def _last_selfattention(self, img, image_feat):
        """The ViT's `get_last_selfattention` (B, heads, P+1, P+1) (src/modules.py:103-104).  A backbone that has the method is
        asked; the convolutional stand-in only builds one when the LHP module will read it (`propagation_strategy == "attn"`):
        softmax similarities of [mean token, patch tokens], channels split over 6 / 12 heads - the ViT's shapes, nothing more."""
        if hasattr(self.model, "get_last_selfattention"):
            return self.model.get_last_selfattention(img)
        if not (getattr(self.cfg, "lhp", False) and getattr(self.cfg, "propagation_strategy", "depth") == "attn"):
            return torch.zeros(1, device=image_feat.device)       # placeholder: only `is None` is ever asked of it
        b, c, h, w = image_feat.shape
        heads = 6 if self.n_feats == 384 else 12
        tok = image_feat.flatten(2).transpose(1, 2)                                   # (B,P,C)
        tok = torch.cat([tok.mean(1, keepdim=True), tok], dim=1).reshape(b, h * w + 1, heads, c // heads).transpose(1, 2)
        return torch.softmax(tok @ tok.transpose(-1, -2) / (c // heads) ** 0.5, dim=-1)

    def forward(self, img, n=1, return_class_feat=False):
        self.model.eval()
        with torch.no_grad():
            assert img.shape[2] % self.patch_size == 0 and img.shape[3] % self.patch_size == 0   # :93-94
            image_feat = self.model(img)
            if return_class_feat:
                return image_feat.mean((2, 3), keepdim=True)
            attn = self._last_selfattention(img, image_feat)
        if self.proj_type is not None:
            # one fused HIP launch: code = cluster1(drop(f)) [+ cluster2(drop(f))] and feats = drop(f) (:122-132; three draws)
            code, feats = run_head(self.cluster1, self.cluster2 if self.proj_type == "nonlinear" else None, image_feat,
                                   self.training, bool(self.cfg.dropout), float(self.dropout.p))
        else:
            code = image_feat
            feats = self.dropout(image_feat) if self.cfg.dropout else image_feat    # :129-137 (identity in eval mode)
        return (feats, code, attn) if self.training else (feats, code)

    supports_deferred_dropout = True      # forward_pair(..., defer_feats_dropout=True) hands back ops.DeferredDropout feats

    def forward_pair(self, img, img_pos, defer_feats_dropout=False):
        """forward(img) and forward(img_pos) of one training step (src/train_segmentation.py:194-212) with the head's two passes in
        one set of launches (run_head_pair): the frozen backbone has no random draws, so the six Dropout2d draws come in the
        reference's order.  Training mode with a projection head only; returns ((feats, code, attn), (feats_pos, code_pos, attn_pos))."""
        if not self.training or self.proj_type is None:
            return self.forward(img), self.forward(img_pos)
        self.model.eval()
        with torch.no_grad():
            assert img.shape[2] % self.patch_size == 0 and img.shape[3] % self.patch_size == 0
            image_feat, image_feat_pos = self.model(img), self.model(img_pos)
            attn, attn_pos = self._last_selfattention(img, image_feat), self._last_selfattention(img_pos, image_feat_pos)
        (code, feats), (code_pos, feats_pos) = run_head_pair(self.cluster1, self.cluster2 if self.proj_type == "nonlinear" else None,
                                                             image_feat, image_feat_pos, True, bool(self.cfg.dropout), float(self.dropout.p),
                                                             None, defer_feats_dropout)
        return (feats, code, attn), (feats_pos, code_pos, attn_pos)


class UnsupervisedSegmenter(nn.Module):
    """LitUnsupervisedSegmenter(n_classes, cfg) without the Lightning plumbing (src/train_segmentation.py:71-158)."""

    def __init__(self, n_classes: int, cfg, net: Optional[nn.Module] = None):
        super().__init__()
        self.cfg, self.n_classes = cfg, n_classes
        dim = n_classes if not cfg.continuous else cfg.dim                          # :78-81
        self.use_depth = bool(cfg.use_depth)
        self.net = net if net is not None else StandInFeaturizer(dim, cfg)           # :99-108 (arch == "dino")
        self.train_cluster_probe = ClusterLookup(dim, n_classes)                      # :110
        self.cluster_probe = ClusterLookup(dim, n_classes + cfg.extra_clusters)       # :112
        self.linear_probe = nn.Conv2d(dim, n_classes, (1, 1))                         # :113
        self.linear_probe_loss_fn = nn.CrossEntropyLoss()                             # :127 (kept for the surface; the step uses the fused HIP loss)
        self.contrastive_corr_loss_fn = ContrastiveCorrelationLoss(cfg)               # :131 (shares cfg: the decay below mutates it)
        for p in self.contrastive_corr_loss_fn.parameters():                          # :136
            p.requires_grad = False
        if getattr(cfg, "lhp", False):                                                 # :82-87
            original = "lhp_original" in str(getattr(cfg, "experiment_name", ""))
            self.lhp_module = OriginalLocalHiddenPositiveProjection(cfg) if original else LocalHiddenPositiveProjection(cfg)
        self.automatic_optimization = False                                           # :139
        self.global_step = 0
        self._optims = None

    def forward(self, x):
        return self.net(x)[1]                                                          # :160-167

    def configure_optimizers(self):                                                    # :537-547
        # as the reference: net_optim steps `self.net.parameters()` only (its decoder is out of scope here, rec_weight = 0) - the
        # LHP projection head is NOT among them, so it keeps its initial weights there and here
        main_params = list(self.net.parameters())
        net_optim = torch.optim.Adam([p for p in main_params if p.requires_grad], lr=self.cfg.lr)
        linear_probe_optim = torch.optim.Adam(list(self.linear_probe.parameters()), lr=5e-3)
        cluster_probe_optim = torch.optim.Adam(list(self.cluster_probe.parameters()), lr=5e-3)
        return net_optim, linear_probe_optim, cluster_probe_optim

    def optimizers(self):
        if self._optims is None:
            self._optims = self.configure_optimizers()
        return self._optims

    def head_parameters(self):
        """The parameters net_optim steps: cluster1 + cluster2."""
        return [p for p in self.net.parameters() if p.requires_grad]

    def all_reduced_parameters(self):
        """Every parameter one of the three optimisers steps - the head (net_optim), the linear probe and the cluster probe.
        Under data parallelism ALL of them must be averaged over the ranks before the optimiser steps (build the GradBucket
        from this list), or the replicas' probes drift apart.  (The LHP head is stepped by no optimiser, as in the reference,
        src/train_segmentation.py:537-547, so it needs no exchange.)"""
        return self.head_parameters() + list(self.linear_probe.parameters()) + list(self.cluster_probe.parameters())

    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int = 0, grad_sync=None):
        """One optimisation step (src/train_segmentation.py:169-462).  `grad_sync`: callable run between backward and the
        optimiser steps (data parallelism: GradBucket pack / all-reduce / unpack); returns (loss, logs)."""
        cfg = self.cfg
        net_optim, linear_probe_optim, cluster_probe_optim = self.optimizers()
        net_optim.zero_grad(); linear_probe_optim.zero_grad(); cluster_probe_optim.zero_grad()        # :174-176
        img, img_pos, label = batch["img"], batch["img_pos"], batch["label"]
        depth = batch["depth"] if self.use_depth else None
        depth_pos = batch["depth_pos"] if self.use_depth else None

        # (both featurizer passes at once where nothing that draws random numbers stands between them in the reference - the LHP
        #  module does - and the featurizer offers it)
        paired = cfg.correspondence_weight > 0 and not getattr(cfg, "lhp", False) and hasattr(self.net, "forward_pair") and self.net.training
        if paired:
            # (on the dense identity grid the Dropout2d of the returned feats is applied by the loss's operand preparation: the
            #  dropped feature tensors are never written - ops.DeferredDropout)
            p = getattr(self.net, "patch_size", None)
            defer = p is not None and getattr(self.net, "supports_deferred_dropout", False) and \
                self.contrastive_corr_loss_fn.takes_deferred_dropout((img.shape[2] // p, img.shape[3] // p))
            (feats, code, attn), (feats_pos, code_pos, _) = self.net.forward_pair(img, img_pos, defer) if defer else \
                self.net.forward_pair(img, img_pos)                                                  # :194-200, :207-212
        else:
            feats, code, attn = self.net(img)                                                        # :194-200
        lhp_code = self.lhp_module(code, depth, img, attn) if getattr(cfg, "lhp", False) else None    # :202-203
        logs: Dict[str, torch.Tensor] = {}
        loss = 0
        if cfg.correspondence_weight > 0:
            if not paired:
                feats_pos, code_pos, _ = self.net(img_pos)                                           # :207-212
            lhp_code_pos = self.lhp_module(code_pos, None) if getattr(cfg, "lhp", False) else None   # :214-215
            salience = batch["mask"].to(torch.float32).squeeze(1) if cfg.use_salience else None      # :233-238
            salience_pos = batch["mask_pos"].to(torch.float32).squeeze(1) if cfg.use_salience else None
            use_depth_term = bool(cfg.depth_feat_correlation_loss)
            d_args = (depth, depth_pos) if use_depth_term else (None, None)                          # :242-292
            out = self.contrastive_corr_loss_fn(feats, feats_pos, salience, salience_pos, code, code_pos, *d_args)
            lhp_out = None
            if getattr(cfg, "lhp", False) and use_depth_term:
                lhp_out = self.contrastive_corr_loss_fn(feats, feats_pos, salience, salience_pos, lhp_code, lhp_code_pos, *d_args)
            total, logs = correspondence_total(cfg, out, lhp_out)                                    # :303-350
            loss = loss + total
        # the legacy decays sit at function-body level in the reference: they run every step, whatever correspondence_weight is
        legacy_decay_step(cfg, self.contrastive_corr_loss_fn.cfg, self.global_step)                  # :356-375 (mutates cfg)

        # probes on the detached code (:421-444)
        detached_code = code.detach().clone()
        # resize to the label resolution + masked cross entropy in one HIP kernel (the reference materialises the logits at label
        # resolution and three masks, :427-434); the cluster probe's similarity / arg-max / loss likewise
        linear_loss = probe_cross_entropy(self.linear_probe(detached_code), label)
        cluster_loss, _ = self.cluster_probe(detached_code, None)
        loss = loss + linear_loss + cluster_loss
        logs.update({"loss/linear": linear_loss.detach(), "loss/cluster": cluster_loss.detach(), "loss/total": loss.detach()})

        loss.backward()                                                                               # :446
        if grad_sync is not None:
            grad_sync()
        net_optim.step(); cluster_probe_optim.step(); linear_probe_optim.step()                       # :447-449
        if cfg.reset_probe_steps is not None and self.global_step == cfg.reset_probe_steps:           # :451-455
            self.linear_probe.reset_parameters()
            self.cluster_probe.reset_parameters()
            self._optims = (net_optim, torch.optim.Adam(list(self.linear_probe.parameters()), lr=5e-3),
                            torch.optim.Adam(list(self.cluster_probe.parameters()), lr=5e-3))
        self.global_step += 1
        return loss.detach(), logs


def default_segmenter_cfg(**over) -> SimpleNamespace:
    """The keys training_step and the featurizer read, with the values of src/configs/local_config.yml."""
    cfg = SimpleNamespace(
        # featurizer / head
        model_type="vit_small", dino_patch_size=8, dino_feat_type="feat", projection_type="nonlinear", dropout=True,
        pretrained_weights=None, continuous=True, dim=70, extra_clusters=0, arch="dino", lr=5e-4, reset_probe_steps=None,
        # loss (hot path)
        feature_samples=11, use_salience=False, depth_sampling="fps", fps_gpu=False, pointwise=True, zero_clamp=True, stabalize=False,
        pos_intra_shift=0.18, pos_inter_shift=0.12, neg_inter_shift=0.46, neg_samples=5,
        depth_feat_correlation_loss=True, depth_feat_shift=0.03,
        # caller
        use_depth=True, use_true_labels=False, correspondence_weight=1.0, pos_inter_weight=0.25, pos_intra_weight=0.67,
        neg_inter_weight=0.63, depth_feat_weight=0.19, rec_weight=0.0, aug_alignment_weight=0.0, crf_weight=0.0, hist_freq=100,
        depth_loss_decay=True, depth_loss_decay_factor=0.6, decay_every_steps=250, fix_depth_feat_shift=False,
        fps_until_step=0, post_fps_samples=11, fps_sample_decay=True, fps_sample_decay_every_steps=1000,
        fps_sample_decay_factor=0.9, fps_min_samples=0, lhp=False, lhp_weight=0.2, lhp_weight_balance=False,
        lhp_depth_weight=0.5)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg
