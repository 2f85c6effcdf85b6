"""Host-side mirror of the reference's operator surface for the correlation loss.

`ContrastiveCorrelationLoss(cfg)` is a drop-in for the reference class of the same name
(src/modules.py:1221-1367): same constructor, a mutable `.cfg` that is re-read on every call
(the caller rewrites `cfg.depth_sampling`, `cfg.feature_samples`, `cfg.depth_feat_shift` between
steps, src/train_segmentation.py:356-375), no parameters, same positional/keyword call signature
and the same 6- / 8-tuple.  All arithmetic runs in the HIP library behind include/depthg_corr.h;
there is no eager/CPU fallback (a missing library or a CPU tensor raises).

Extra, build-side cfg keys (all optional, read with getattr):
    dg_outputs      "full" (default): tuple elements 1,3,4,5,7 are the reference's un-reduced
                    (.., S,S,S,S) tensors, written by dg_corr_materialize;
                    "reduced": they are 1-element tensors holding the mean (so the caller's
                    `.mean()` / logging keeps working) and nothing of size (B,P,P) ever reaches HBM.
    dg_dense_grid   True: with feature_samples == h == w use the identity grid for coords1 and
                    coords2 (SURVEY.md section 8(d) dense runs) instead of torch.rand.
After a call, `.scalars` is the fused fp32 output vector (DG_OUT_* order: the four loss means, the four cd means, the
weighted total) with its grad_fn and `.total` its last element = the term `training_step` adds to its loss
(src/train_segmentation.py:330-349, weights read from cfg), so `loss_fn.total.backward()` needs no further torch ops.
"""
import torch
import torch.nn as nn

from . import ops


SMALL_GRID_POSITIONS = 160      # sample grids up to here run the fused small-grid kernel (dg_small.hip), identity coordinates included


def super_perm(size: int, device) -> torch.Tensor:
    """randperm with fixed points bumped by one, modulo size (src/modules.py:1184-1188; quirk Q6)."""
    perm = torch.randperm(size, device=device, dtype=torch.long)
    perm = torch.where(perm == torch.arange(size, device=device), perm + 1, perm)
    return perm % size


def super_perms(count: int, size: int, device) -> torch.Tensor:
    """`count` independent super_perm draws as one (count, size) tensor with ONE sort launch: argsort of iid uniforms
    is a uniform random permutation, like randperm; then the same fixed-point bump (src/modules.py:1186-1188)."""
    if count == 0:
        return torch.zeros(0, size, dtype=torch.long, device=device)
    perm = torch.argsort(torch.rand(count, size, device=device), dim=1)
    ar = torch.arange(size, device=device).unsqueeze(0)
    return torch.where(perm == ar, perm + 1, perm) % size


def identity_coords(b: int, s: int, device) -> torch.Tensor:
    """coords such that sample() reads every pixel of an s x s map exactly once (transposed, quirk Q3)."""
    lin = torch.linspace(-1.0, 1.0, s, device=device)
    c = torch.empty(b, s, s, 2, device=device)
    c[..., 0] = lin.view(1, 1, s)
    c[..., 1] = lin.view(1, s, 1)
    return c


class _CorrLossFunction(torch.autograd.Function):
    """out[DG_OUT_COUNT] = fused loss scalars (+ weighted total); backward = dg_corr_backward with the upstream of that vector."""

    @staticmethod
    def forward(ctx, orig_code, orig_code_pos, orig_feats, orig_feats_pos, depth, coords1, coords2, perms, desc, holder):
        ws = ops.alloc_workspace(desc, orig_feats.device)
        fk = holder.get("feat_keep")
        if holder.get("feat_inv") is not None:     # one channel chunk of wider maps on sampled coordinates (forward_with: `wide`)
            out = ops.corr_forward_extnorm(desc, orig_feats, orig_feats_pos, orig_code, orig_code_pos, depth, coords1, coords2,
                                           perms, holder["feat_inv"], ws)
        elif fk is not None:               # Dropout2d of the feature maps applied inside the operand preparation (identity grid)
            drew = perms is None
            out, perms = ops.corr_forward_masked(desc, orig_feats, orig_feats_pos, orig_code, orig_code_pos, depth, coords1, coords2,
                                                 perms, ws, holder.get("draw_state"), fk[0], fk[1], fk[2])
            if drew:
                holder["perms"] = perms
        elif perms is None:                # the forward draws the negatives itself (holder["draw_state"]: device generator or None)
            out, perms = ops.corr_forward_draw(desc, orig_feats, orig_feats_pos, orig_code, orig_code_pos, depth, coords1, coords2,
                                               ws, state=holder.get("draw_state"))
            holder["perms"] = perms
        else:
            out = ops.corr_forward(desc, orig_feats, orig_feats_pos, orig_code, orig_code_pos, depth, coords1, coords2,
                                   perms, ws)
        holder["workspace"] = ws
        ctx.desc, ctx.ws, ctx.shape = desc, ws, tuple(orig_code.shape)
        ctx.save_for_backward(coords1, coords2, perms)
        # second output: the weighted total as its own autograd output (a view of element DG_OUT_TOTAL), so that
        # `total.backward()` reaches backward() with a scalar instead of going through a select-backward (zeros + copy)
        ctx.set_materialize_grads(False)
        return out, out[ops._lib.DG_OUT_TOTAL]

    @staticmethod
    def backward(ctx, gout, gtotal):
        nothing = (None,) * 10
        if gout is None and gtotal is None:
            return nothing
        coords1, coords2, perms = ctx.saved_tensors
        if not (ctx.desc.flags & ops._lib.DG_NEED_GRAD):
            raise RuntimeError("depthg_amd: backward called on a forward that ran without gradient pieces")
        if gout is None:
            gt = gtotal.to(torch.float32).contiguous()
            g_code, g_code_pos = ops.corr_backward_total(ctx.desc, gt, coords1, coords2, perms, ctx.ws, ctx.shape)
        else:
            gs = gout.to(torch.float32).contiguous()      # [DG_OUT_COUNT]: entries 0..3 and DG_OUT_TOTAL are used
            if gtotal is not None:
                gs = gs.clone()
                gs[ops._lib.DG_OUT_TOTAL] += gtotal.to(torch.float32)
            g_code, g_code_pos = ops.corr_backward(ctx.desc, gs, coords1, coords2, perms, ctx.ws, ctx.shape)
        return (g_code, g_code_pos) + nothing[2:]


def _rand_coords(shape, device):
    """`torch.rand(shape) * 2 - 1` of the reference (src/modules.py:1310-1321) as ONE launch: `uniform_(-1, 1)` consumes the same
    Philox draws and forms u * 2 + (-1) - bit-identical on the GPU (scripts/rand_eq.py), two 5-us elementwise launches less per
    coordinate set."""
    return torch.empty(shape, device=device).uniform_(-1, 1)


class ContrastiveCorrelationLoss(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self._ident_cache = (None, None)
        self._perm_state = None            # device generator state of the negatives' permutations (cfg.dg_graph_safe)

    def _total_weights(self, depth_term):
        """Weights of (intra, inter, neg, depth) loss means in the caller's total, src/train_segmentation.py:325-349."""
        cfg = self.cfg
        g = lambda k: float(getattr(cfg, k, 0.0))
        balance = g("lhp_weight") if (depth_term and getattr(cfg, "lhp", False) and getattr(cfg, "lhp_weight_balance", False)) else 0.0
        scale = g("correspondence_weight") - balance
        return (g("pos_intra_weight") * scale, g("pos_inter_weight") * scale, g("neg_inter_weight") * scale,
                g("depth_feat_weight") * scale if depth_term else 0.0)

    def takes_deferred_dropout(self, feat_hw, code_hw=None) -> bool:
        """True when a call with feature maps of size `feat_hw` runs the identity grid (_draw_coords), i.e. when the Dropout2d of the
        feature maps can be applied inside the operand preparation (ops.DeferredDropout) instead of by the featurizer."""
        cfg = self.cfg
        if getattr(cfg, "use_salience", False) or cfg.depth_sampling in ("simple", "fps", "fps_depth_feat"):
            return False
        S = int(cfg.feature_samples)
        return bool(getattr(cfg, "dg_dense_grid", False)) and (code_hw is None or tuple(code_hw) == tuple(feat_hw)) and \
            S == feat_hw[0] == feat_hw[1]

    # -- coordinate selection, src/modules.py:1287-1321 -------------------------------------------
    def _draw_coords(self, orig_feats, orig_feats_pos, orig_salience, orig_salience_pos, depth, depth_pos, same_maps=True):
        cfg = self.cfg
        B, S = orig_feats.shape[0], int(cfg.feature_samples)
        dev = orig_feats.device
        coord_shape = [B, S, S, 2]
        if getattr(cfg, "use_salience", False):          # src/modules.py:1290-1297
            c1_nonzero = ops.salience_coords(orig_salience, S)
            c2_nonzero = ops.salience_coords(orig_salience_pos, S)
            c1_reg = _rand_coords(coord_shape, dev)
            c2_reg = _rand_coords(coord_shape, dev)
            mask = (torch.rand(coord_shape[:-1], device=dev) > .1).unsqueeze(-1).to(torch.float32)
            return c1_nonzero * mask + c1_reg * (1 - mask), c2_nonzero * mask + c2_reg * (1 - mask), False
        mode = cfg.depth_sampling
        if mode == "simple":                             # src/modules.py:1299-1302: S x 1 grids, (B,S,1,2)
            if depth is None or depth_pos is None:
                raise AttributeError("depth_sampling='simple' needs depth and depth_pos")
            c1 = ops.simple_depth_coords(depth, orig_feats.shape[-2:], S)
            c2 = ops.simple_depth_coords(depth_pos, orig_feats_pos.shape[-2:], S)
            return c1, c2, False
        if mode in ("fps", "fps_depth_feat"):   # 'fps_depth_feat' == 'fps' in the reference (quirk Q13)
            if depth is None or depth_pos is None:
                raise AttributeError("depth_sampling='fps' needs depth and depth_pos (the reference fails on None too, quirk Q8)")
            if depth.shape == depth_pos.shape and orig_feats.shape[-2:] == orig_feats_pos.shape[-2:]:
                # one launch for both maps: the sampler is a sequential per-image loop (one block per image), so the two
                # calls of the reference (src/modules.py:1304-1308) simply run side by side on twice as many CUs
                both = ops.fps_coords_pair(depth, depth_pos, orig_feats.shape[-2:], S)
                c1, c2 = both[:B], both[B:]
            else:
                c1 = ops.fps_coords(depth, orig_feats.shape[-2:], S)
                c2 = ops.fps_coords(depth_pos, orig_feats_pos.shape[-2:], S)
            assert tuple(c1.shape) == tuple(c2.shape) == tuple(coord_shape), f"{c1.shape} != {c2.shape} != {coord_shape}"
            return c1, c2, False
        if getattr(cfg, "dg_dense_grid", False) and same_maps and S == orig_feats.shape[-2] == orig_feats.shape[-1]:
            key = (B, S, str(dev))
            if self._ident_cache[0] != key:          # constant tensor: built once per (B, S, device)
                self._ident_cache = (key, identity_coords(B, S, dev))
            c = self._ident_cache[1]
            return c, c, True
        if getattr(cfg, "dg_graph_safe", False):
            # the step is recorded in a hipGraph: both coordinate sets from the device-resident generator the negatives' batch maps
            # use, one launch (torch's generator inside a graph: one launch per tensor and two 64-bit fills per replay)
            c1, c2 = ops.rand_coords_state(self._graph_state(dev), coord_shape)
            return c1, c2, False
        c1 = _rand_coords(coord_shape, dev)
        c2 = _rand_coords(coord_shape, dev)
        return c1, c2, False

    def _graph_state(self, dev):
        """Generator state on the device (hipGraph-capturable step: nothing about a draw is baked into a launch)."""
        if self._perm_state is None or self._perm_state.device != dev:
            self._perm_state = ops.new_perm_state(dev)
        return self._perm_state

    @staticmethod
    def _check_maps(orig_feats, orig_feats_pos, orig_code, orig_code_pos, depth):
        """The C side receives raw pointers: everything it assumes about the four maps is checked here.  The reference's
        `sample()` (src/modules.py:822-825) takes normalised coordinates, so the code maps may have another resolution than the
        feature maps (FeaturePyramidNet: low_res_feats (B,2048,7,7), code (B,dim,56,56), src/modules.py:732-766); what torch
        itself would refuse there - another batch size (grid_sample), another channel count between the two operands of a
        correlation (einsum), tensors on different devices - raises RuntimeError here as well."""
        maps = (("orig_feats", orig_feats), ("orig_feats_pos", orig_feats_pos), ("orig_code", orig_code),
                ("orig_code_pos", orig_code_pos))
        for name, t in maps:
            if not isinstance(t, torch.Tensor) or t.dim() != 4:
                raise ValueError(f"depthg_amd: `{name}` must be a 4-D (B,K,h,w) tensor, got "
                                 f"{tuple(t.shape) if isinstance(t, torch.Tensor) else type(t).__name__}")
            if not t.is_floating_point():
                raise ValueError(f"depthg_amd: `{name}` must be a floating-point tensor, got {t.dtype}")
        dev, B = orig_feats.device, orig_feats.shape[0]
        for name, t in maps[1:] + ((("depth", depth),) if depth is not None else ()):
            if t.device != dev:
                raise RuntimeError(f"depthg_amd: `{name}` lives on {t.device}, `orig_feats` on {dev}: all inputs of one call must "
                                   f"share a device")
            if t.shape[0] != B:
                raise RuntimeError(f"depthg_amd: `{name}` has batch size {t.shape[0]}, `orig_feats` has {B} (the reference's "
                                   f"grid_sample / einsum refuse that too)")
        if tuple(orig_feats_pos.shape) != tuple(orig_feats.shape):
            raise RuntimeError(f"depthg_amd: orig_feats_pos {tuple(orig_feats_pos.shape)} must have the shape of orig_feats "
                               f"{tuple(orig_feats.shape)} (one featurizer produces both)")
        if tuple(orig_code_pos.shape) != tuple(orig_code.shape):
            raise RuntimeError(f"depthg_amd: orig_code_pos {tuple(orig_code_pos.shape)} must have the shape of orig_code "
                               f"{tuple(orig_code.shape)} (one head produces both)")
        if depth is not None and (depth.dim() != 4 or depth.shape[1] != 1):
            raise ValueError(f"depthg_amd: `depth` must be (B,1,H,W), got {tuple(depth.shape)}")

    def forward(self, orig_feats, orig_feats_pos, orig_salience, orig_salience_pos, orig_code, orig_code_pos,
                depth=None, depth_pos=None):
        orig_feats, orig_feats_pos, feat_keep = self._unwrap_deferred(orig_feats, orig_feats_pos)
        self._check_maps(orig_feats, orig_feats_pos, orig_code, orig_code_pos, depth)
        if depth_pos is not None and depth is not None and (depth_pos.device != depth.device or depth_pos.shape[0] != depth.shape[0]):
            raise RuntimeError(f"depthg_amd: depth_pos {tuple(depth_pos.shape)} on {depth_pos.device} does not match depth "
                               f"{tuple(depth.shape)} on {depth.device}")
        coords1, coords2, shared = self._draw_coords(orig_feats, orig_feats_pos, orig_salience, orig_salience_pos,
                                                     depth, depth_pos, same_maps=orig_code.shape[-2:] == orig_feats.shape[-2:])
        state = None
        if getattr(self.cfg, "dg_graph_safe", False):
            state = self._graph_state(orig_feats.device)
        # the negatives' batch maps (super_perm, src/modules.py:1340-1342) are drawn by the forward itself: perms=None
        # (seed from torch's CPU generator unless the device generator is in use); `shared` is only ever set together with the
        # identity grid drawn above
        return self.forward_with(orig_feats, orig_feats_pos, orig_code, orig_code_pos, depth, coords1, coords2, None,
                                 shared_coords=shared, identity_grid=shared, draw_state=state, _checked=True, feat_keep=feat_keep)

    @staticmethod
    def _unwrap_deferred(orig_feats, orig_feats_pos):
        """ops.DeferredDropout in place of a feature map (the head's forward_pair(..., defer_feats_dropout=True)): the un-dropped
        maps and (keep, keep_pos, scale) for dg_corr_forward_masked - or None when neither is deferred.  Two different scales
        cannot ride in one call: the second map is then materialised here."""
        da = orig_feats if isinstance(orig_feats, ops.DeferredDropout) else None
        db = orig_feats_pos if isinstance(orig_feats_pos, ops.DeferredDropout) else None
        if da is None and db is None:
            return orig_feats, orig_feats_pos, None
        if da is not None and db is not None and da.scale != db.scale:
            orig_feats_pos, db = db.materialize(), None
        scale = da.scale if da is not None else db.scale
        f32 = lambda k: k.detach().to(torch.float32).contiguous()
        return (da.feats if da is not None else orig_feats, db.feats if db is not None else orig_feats_pos,
                (f32(da.keep) if da is not None else None, f32(db.keep) if db is not None else None, scale))

    # -- everything after the RNG draws (explicit coords / perms: parity tests, DP shards) ----------
    def forward_with(self, orig_feats, orig_feats_pos, orig_code, orig_code_pos, depth, coords1, coords2, perms,
                     shared_coords=False, identity_grid=False, draw_state=None, _checked=False, feat_keep=None):
        cfg = self.cfg
        if feat_keep is None:
            orig_feats, orig_feats_pos, feat_keep = self._unwrap_deferred(orig_feats, orig_feats_pos)
        if not _checked:
            self._check_maps(orig_feats, orig_feats_pos, orig_code, orig_code_pos, depth)
        if identity_grid and tuple(orig_code.shape[-2:]) != tuple(orig_feats.shape[-2:]):
            raise ValueError(f"depthg_amd: the identity grid needs code maps of the feature maps' size, got "
                             f"{orig_code.shape[-2]}x{orig_code.shape[-1]} against {orig_feats.shape[-2]}x{orig_feats.shape[-1]}")
        if identity_grid and int(cfg.feature_samples) ** 2 <= SMALL_GRID_POSITIONS and not getattr(cfg, "dg_small_identity_blobs", False):
            # Dense grids of at most 160 positions (7 x 7 ... 12 x 12) do NOT take the identity-grid launches: the identity
            # coordinates go through the sampler like any others (what the reference does with them) and the fused small-grid kernel
            # runs, whose clamp masks are exact.  The blob kernels' fp16 cd decides the mask there, and on so few positions a handful of
            # flipped elements is percents of a gradient (round 5's randomised sweep, seed 5133: 7 x 7, d/d code_pos 4.1e-2 rel-L2).
            # (cfg.dg_small_identity_blobs: the old route, kept for A/B runs and the C-ABI tests of DG_IDENTITY_GRID at small P.)
            identity_grid = False
        if feat_keep is not None:
            ka, kb, kscale = feat_keep
            for name, k in (("orig_feats", ka), ("orig_feats_pos", kb)):
                if k is not None and (tuple(k.shape) != tuple(orig_feats.shape[:2]) or k.device != orig_feats.device):
                    raise RuntimeError(f"depthg_amd: keep flags of `{name}` are {tuple(k.shape)} on {k.device}; the maps are "
                                       f"{tuple(orig_feats.shape)} on {orig_feats.device}")
            if not identity_grid:
                # sampled coordinates: the samplers read the maps themselves - the dropped tensors are formed here (same values)
                if ka is not None:
                    orig_feats = orig_feats * (ka * kscale)[:, :, None, None]
                if kb is not None:
                    orig_feats_pos = orig_feats_pos * (kb * kscale)[:, :, None, None]
                feat_keep = None
        B, C, h, w = orig_feats.shape
        D, hc, wc = orig_code.shape[1:]
        same_maps = (hc, wc) == (h, w)
        S, N = int(cfg.feature_samples), int(cfg.neg_samples)
        if tuple(coords1.shape) != tuple(coords2.shape) or tuple(coords1.shape) not in ((B, S, S, 2), (B, S, 1, 2)):
            raise ValueError(f"depthg_amd: coords must both be (B,S,S,2) or (B,S,1,2) with B={B}, S={S}; got "
                             f"{tuple(coords1.shape)} and {tuple(coords2.shape)}")
        depth_term = bool(cfg.depth_feat_correlation_loss)
        if depth_term and depth is None:
            raise AttributeError("depth_feat_correlation_loss=True needs `depth` (reference: interpolate(None) raises)")
        dev = orig_feats.device
        feats = ops._f32c(orig_feats, "orig_feats")
        feats_pos = ops._f32c(orig_feats_pos, "orig_feats_pos")
        depth_c = ops._f32c(depth, "depth") if depth_term else None
        coords1 = ops._f32c(coords1, "coords1")
        coords2 = ops._f32c(coords2, "coords2")
        if coords1.device != dev or coords2.device != dev:
            raise RuntimeError(f"depthg_amd: coords live on {coords1.device} / {coords2.device}, the maps on {dev}")
        line_grid = coords1.shape[2] == 1 and S != 1       # S x 1 grid of depth_sampling='simple'
        if perms is None:
            perms_t = None                 # drawn inside the forward (dg_corr_forward_draw)
        elif isinstance(perms, (list, tuple)):
            perms_t = torch.stack([p.to(device=dev, dtype=torch.long) for p in perms]) if N > 0 else \
                torch.zeros(0, B, dtype=torch.long, device=dev)
        else:
            perms_t = perms.to(device=dev, dtype=torch.long)
        if perms_t is not None:
            perms_t = perms_t.contiguous()
            assert perms_t.shape == (N, B), f"perms shape {tuple(perms_t.shape)} != {(N, B)}"
        need_grad = torch.is_grad_enabled() and (orig_code.requires_grad or orig_code_pos.requires_grad)
        code_in = orig_code if orig_code.dtype == torch.float32 else orig_code.float()
        code_pos_in = orig_code_pos if orig_code_pos.dtype == torch.float32 else orig_code_pos.float()
        code_in, code_pos_in = code_in.contiguous(), code_pos_in.contiguous()
        all_shifts = (cfg.pos_intra_shift, cfg.pos_inter_shift, cfg.neg_inter_shift, cfg.depth_feat_shift if depth_term else 0.0)

        def run(f_a, f_b, perms_in, first=True, unit=False, keep=feat_keep, feat_inv=None):
            """one launch set of the C ABI on feature maps of the width the operand kernels hold (`first`: the recipe's shifts and
            depth term; else a further channel chunk of a wider map: zero shifts, no depth term - see below)"""
            dt = depth_term and first
            desc_ = ops.make_desc(B, f_a.shape[1], D, h, w, S, N, pointwise=bool(cfg.pointwise), zero_clamp=bool(cfg.zero_clamp),
                                  stabalize=bool(cfg.stabalize), depth_term=dt, need_grad=need_grad,
                                  shared_coords=bool(shared_coords),
                                  shifts=all_shifts if first else (0.0, 0.0, 0.0, 0.0),
                                  depth_hw=tuple(depth_c.shape[-2:]) if (depth_c is not None and dt) else (0, 0),
                                  identity_grid=bool(identity_grid), weights=self._total_weights(dt),
                                  line_grid=line_grid, code_hw=None if same_maps else (hc, wc),
                                  exact_masks=bool(getattr(cfg, "dg_exact_masks", False)), feats_unit=unit)
            holder_ = {"draw_state": draw_state, "feat_keep": keep, "feat_inv": feat_inv}
            out_, total_ = _CorrLossFunction.apply(code_in, code_pos_in, f_a, f_b, depth_c if dt else None,
                                                   coords1, coords2, perms_in, desc_, holder_)
            return out_, total_, desc_, holder_["workspace"], (holder_["perms"] if perms_in is None else perms_in)

        wide = identity_grid and C > ops.BLOB_MAX_C
        P_pos = int(coords1.shape[1]) * int(coords1.shape[2])
        wide_sampled = (not identity_grid) and C > ops.BLOB_MAX_C and P_pos > SMALL_GRID_POSITIONS
        if wide_sampled:
            # ... and on SAMPLED coordinates above 160 positions: the reference normalises behind sample(), so the norm of every sampled
            # vector is formed over all channel chunks first (dg_sampled_sumsq per operand: feats at coords1, feats_pos at coords2 and,
            # with coordinates per image, feats through every negative's batch map at coords2), then each chunk runs with those norms
            # (dg_corr_forward_extnorm) - shifts, depth term and the sums as on the dense grid below
            if feat_keep is not None:
                raise RuntimeError("depthg_amd: deferred feature dropout is the identity grid's")       # (forward_with formed the tensors above)
            if perms_t is None:
                perms_t = super_perms(N, B, dev) if N > 0 else torch.zeros(0, B, dtype=torch.long, device=dev)
            nops = 2 if (shared_coords or N == 0) else 2 + N
            nch = (C + ops.BLOB_MAX_C - 1) // ops.BLOB_MAX_C
            chunk_c = ((C + nch - 1) // nch + 7) // 8 * 8
            fa = [feats[:, k:k + chunk_c].contiguous() for k in range(0, C, chunk_c)]
            fb = [feats_pos[:, k:k + chunk_c].contiguous() for k in range(0, C, chunk_c)]
            sumsq = torch.empty(nops, B, P_pos, device=dev, dtype=torch.float32)
            for k in range(len(fa)):
                ops.sampled_sumsq(fa[k], coords1, None, sumsq[0], k > 0)
                ops.sampled_sumsq(fb[k], coords2, None, sumsq[1], k > 0)
                for j in range(2, nops):
                    ops.sampled_sumsq(fa[k], coords2, perms_t[j - 2], sumsq[j], k > 0)     # (the negatives are orig_feats[perm] at coords2, src/modules.py:1341-1345)
            feat_inv = (1.0 / sumsq.sqrt().clamp_min(1e-10)).contiguous()
            chunks = []
            for k in range(len(fa)):
                o_k, t_k, desc_k, ws_k, perms_t = run(fa[k], fb[k], perms_t, first=(k == 0), keep=None, feat_inv=feat_inv)
                chunks.append((o_k, t_k, desc_k, ws_k))
            lossmask = torch.zeros(ops._lib.DG_OUT_COUNT, device=dev)
            lossmask[[0, 1, 2, ops._lib.DG_OUT_TOTAL]] = 1.0
            out, total = chunks[0][0], chunks[0][1]
            for o_k, t_k, _, _ in chunks[1:]:
                out = out + o_k * lossmask
                total = total + t_k
            desc, ws = chunks[0][2], chunks[0][3]
        elif not wide:
            out, total, desc, ws, perms_t = run(feats, feats_pos, perms_t)
            chunks = None
        else:
            # Feature maps wider than the operand kernels hold, on the dense identity grid: the loss is LINEAR in the feature
            # correlation fd = sum over channels (src/modules.py:797-809; helper(), :1231-1254: the centering of fd, the shift and
            # -clamp(cd) * (fd - shift)), and on this grid sample() is a transposition, so norm() can run in front of it.  The maps are
            # normalised over all C channels once (dg_normalize_split) and evaluated chunk by chunk (DG_FEATS_UNIT): the first chunk
            # with the recipe's shifts and depth term, the others with zero shifts and without it; the loss means and the code
            # gradients add up, the cd means (and everything of the depth term) are the first chunk's.
            if feat_keep is not None:          # (deferred Dropout2d: formed here, in front of the normalisation - same values)
                ka, kb, kscale = feat_keep
                if ka is not None:
                    feats = feats * (ka * kscale)[:, :, None, None]
                if kb is not None:
                    feats_pos = feats_pos * (kb * kscale)[:, :, None, None]
            nch = (C + ops.BLOB_MAX_C - 1) // ops.BLOB_MAX_C
            chunk_c = ((C + nch - 1) // nch + 7) // 8 * 8
            fa, fb = ops.normalize_split(feats, chunk_c), ops.normalize_split(feats_pos, chunk_c)
            chunks = []
            for k in range(len(fa)):
                o_k, t_k, desc_k, ws_k, perms_t = run(fa[k], fb[k], perms_t, first=(k == 0), unit=True, keep=None)
                chunks.append((o_k, t_k, desc_k, ws_k))
            lossmask = torch.zeros(ops._lib.DG_OUT_COUNT, device=dev)
            lossmask[[0, 1, 2, ops._lib.DG_OUT_TOTAL]] = 1.0
            out, total = chunks[0][0], chunks[0][1]
            for o_k, t_k, _, _ in chunks[1:]:
                out = out + o_k * lossmask
                total = total + t_k
            desc, ws = chunks[0][2], chunks[0][3]
        d = self.__dict__                      # plain attributes: nn.Module.__setattr__ costs microseconds per assignment
        d["last_scalars"] = out.detach()
        d["scalars"] = out                     # the fused output vector with its grad_fn (DG_OUT_* order)
        d["total"] = total                     # weighted total of the loss means (src/train_segmentation.py:330-349)
        d["last_call"] = (desc, perms_t, ws)   # measurement aid (bench.py re-launches the fused kernel alone)

        mode = getattr(cfg, "dg_outputs", "full")
        if mode == "reduced":
            res = (out[0], out[4:5].detach(), out[1], out[5:6].detach(), out[2:3], out[6:7].detach())
            if depth_term:
                res = res + (out[3], out[7:8].detach())
            return res
        if mode != "full":
            raise ValueError(f"cfg.dg_outputs must be 'full' or 'reduced', got {mode!r}")
        # reference-shaped outputs (src/modules.py:1352-1367); cd tensors carry no gradient (the caller only logs them)
        # (on the shared dense grid the negatives' operands are the anchors' operand read through the batch maps: the maps go along;
        #  a 28 x 28 tensor is 78.7 MB per pair-set at B = 32 - the reference materialises them on every step, here only when asked)
        intra_cd, _ = ops.corr_materialize(desc, 0, ws)
        inter_cd, _ = ops.corr_materialize(desc, 1, ws)
        neg_cd, neg_loss = [], []
        for k in range(N):
            c, l = ops.corr_materialize(desc, 2 + k, ws, want_cd=True, want_loss=True, perms=perms_t)
            if chunks is not None:             # (a wide map in channel chunks: the un-reduced loss is the sum of the chunks' too)
                for _, _, desc_j, ws_j in chunks[1:]:
                    l = l + ops.corr_materialize(desc_j, 2 + k, ws_j, want_cd=False, want_loss=True, perms=perms_t)[1]
            neg_cd.append(c)
            neg_loss.append(l)
        if N > 0:
            neg_cd_t = torch.cat(neg_cd, dim=0)
            # un-reduced negative loss: values from the kernel; gradient routed through its mean (exact for the
            # uniform upstreams that .mean()/.sum() produce, which is how the caller consumes it,
            # src/train_segmentation.py:303)
            neg_loss_t = torch.cat(neg_loss, dim=0) + (out[2] - out[2].detach())
        else:
            sh = 1 if line_grid else S
            neg_cd_t = torch.zeros(0, sh, S, sh, S, device=dev)
            neg_loss_t = neg_cd_t.clone()
        res = (out[0], intra_cd, out[1], inter_cd, neg_loss_t, neg_cd_t)
        if depth_term:
            dd, _ = ops.corr_materialize(desc, -1, ws)
            res = res + (out[3], dd)
        return res
