"""Tensor-level wrappers over the C ABI (raw device pointers + the current HIP stream).
torch is used for device memory and streams only."""
import ctypes
import os

import torch

from . import _lib
from ._lib import CorrDesc


# Test hook (tests/test_gpu_poison.py, DG_POISON=1): every buffer this layer hands to the library - the workspace and all
# outputs - is pre-filled with 0xFF bytes (a NaN pattern for fp32/fp16, -1 for integers), so that a kernel that reads a
# byte it (or an earlier kernel of the call) has not written shows up as NaN instead of depending on recycled memory.
POISON = os.environ.get("DG_POISON", "") not in ("", "0")


def _empty(shape, dtype, device):
    t = torch.empty(shape, dtype=dtype, device=device)
    if POISON and t.numel():
        t.reshape(-1).view(torch.uint8).fill_(0xFF)
    return t


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(device):
    """Current stream of `device`, as the C ABI wants it.  The library launches on the CURRENT HIP device and keys its
    kernel-attribute cache on it: a call with tensors of another device would run on a foreign stream - refuse it."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"depthg_amd: tensors must live on the GPU (got {device}); there is no CPU path")
    cur = torch.cuda.current_device()
    if device.index is not None and device.index != cur:
        raise RuntimeError(f"depthg_amd: tensors live on cuda:{device.index} but the current device is cuda:{cur}; "
                           f"call under `with torch.cuda.device({device.index}):` (one process per GPU sets it once)")
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _f32c(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"depthg_amd: `{name}` must live on the GPU (got {t.device}); there is no CPU path")
    return t.detach().to(torch.float32).contiguous()


def make_desc(B, C, D, h, w, S, n_neg, *, pointwise, zero_clamp, stabalize, depth_term, need_grad, shared_coords,
              shifts, depth_hw=(0, 0), identity_grid=False, weights=(0.0, 0.0, 0.0, 0.0), line_grid=False, code_hw=None,
              exact_masks=False, feats_unit=False):
    """dg_corr_desc.  (h, w): the feature maps; code_hw: the code maps' size when it differs (None: the same)."""
    flags = 0
    flags |= _lib.DG_POINTWISE if pointwise else 0
    flags |= _lib.DG_ZERO_CLAMP if zero_clamp else 0
    flags |= _lib.DG_STABALIZE if stabalize else 0
    flags |= _lib.DG_DEPTH_TERM if depth_term else 0
    flags |= _lib.DG_NEED_GRAD if need_grad else 0
    flags |= _lib.DG_SHARED_COORDS if shared_coords else 0
    flags |= _lib.DG_IDENTITY_GRID if identity_grid else 0
    flags |= _lib.DG_LINE_GRID if line_grid else 0
    flags |= _lib.DG_EXACT_MASKS if exact_masks else 0
    flags |= _lib.DG_FEATS_UNIT if feats_unit else 0
    ch, cw = (int(code_hw[0]), int(code_hw[1])) if code_hw is not None and tuple(code_hw) != (h, w) else (0, 0)
    return CorrDesc(B, C, D, h, w, S, n_neg, int(depth_hw[0]), int(depth_hw[1]), flags,
                    float(shifts[0]), float(shifts[1]), float(shifts[2]), float(shifts[3]),
                    float(weights[0]), float(weights[1]), float(weights[2]), float(weights[3]), ch, cw)


BLOB_MAX_C = 768          # feature channels the blob kernels (grids above 160 positions, the identity grid) hold per call


def normalize_split(feats, chunk_c):
    """norm() of the reference over all channels of (B,C,h,w), returned as contiguous channel chunks of `chunk_c` (the last: the
    rest) - the operands of DG_FEATS_UNIT calls (dg_normalize_split)."""
    feats = _f32c(feats, "feats")
    B, C, h, w = feats.shape
    n = (C + chunk_c - 1) // chunk_c
    outs = [_empty((B, min(chunk_c, C - k * chunk_c), h, w), torch.float32, feats.device) for k in range(n)]
    ptrs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    _lib.check(_lib.load().dg_normalize_split(B, C, h, w, _ptr(feats), n, chunk_c, ptrs, _stream(feats.device)), "dg_normalize_split")
    return outs


def sampled_sumsq(feats_chunk, coords, srcidx, out, accumulate):
    """out[n][p] (+)= sum over the chunk's channels of sample(feats_chunk[srcidx[n]], coords[n])^2 (dg_sampled_sumsq)."""
    B, C, h, w = feats_chunk.shape
    S, line = coords.shape[1], coords.shape[2] == 1 and coords.shape[1] != 1
    _lib.check(_lib.load().dg_sampled_sumsq(B, C, h, w, S, 1 if line else 0, _ptr(feats_chunk), _ptr(coords), _ptr(srcidx),
                                            1 if accumulate else 0, _ptr(out), _stream(feats_chunk.device)), "dg_sampled_sumsq")


def corr_forward_extnorm(desc, feats, feats_pos, code, code_pos, depth, coords1, coords2, perms, feat_inv, workspace):
    """dg_corr_forward on ONE channel chunk of wider feature maps, normalised by `feat_inv` (nops, B, P): 1 / the norm of the whole
    sampled vector (sampled coordinates above 160 positions; see include/depthg_corr.h)."""
    lib = _lib.load()
    dev = feats.device
    out = _empty(_lib.DG_OUT_COUNT, torch.float32, dev)
    rc = lib.dg_corr_forward_extnorm(ctypes.byref(desc), _ptr(feats), _ptr(feats_pos), _ptr(code), _ptr(code_pos), _ptr(depth),
                                     _ptr(coords1), _ptr(coords2), _ptr(perms), _ptr(feat_inv), _ptr(out), _ptr(workspace),
                                     workspace.numel(), _stream(dev))
    _lib.check(rc, "dg_corr_forward_extnorm")
    return out


def workspace_bytes(desc):
    n = _lib.load().dg_corr_workspace_bytes(ctypes.byref(desc))
    if n == 0:
        _lib.check(-1, "dg_corr_workspace_bytes")
    return n


def alloc_workspace(desc, device):
    return _empty(workspace_bytes(desc), torch.uint8, device)


def corr_forward(desc, feats, feats_pos, code, code_pos, depth, coords1, coords2, perms, workspace):
    """Returns fp32 [DG_OUT_COUNT] device tensor (order: DG_OUT_* of include/depthg_corr.h)."""
    lib = _lib.load()
    dev = feats.device
    out = _empty(_lib.DG_OUT_COUNT, torch.float32, dev)
    rc = lib.dg_corr_forward(ctypes.byref(desc), _ptr(feats), _ptr(feats_pos), _ptr(code), _ptr(code_pos), _ptr(depth),
                             _ptr(coords1), _ptr(coords2), _ptr(perms), _ptr(out), _ptr(workspace),
                             workspace.numel(), _stream(dev))
    _lib.check(rc, "dg_corr_forward")
    return out


def corr_forward_draw(desc, feats, feats_pos, code, code_pos, depth, coords1, coords2, workspace, state=None):
    """dg_corr_forward_draw: the forward draws the negatives' batch maps itself (on the identity grid inside its first launch).
    Returns (out, perms): perms (n_neg, B) int64 is what backward / materialize need.  `state` (new_perm_state) = device-resident
    generator (hipGraph-safe); without it the seed comes from torch's CPU generator, as in super_perms()."""
    lib = _lib.load()
    dev = feats.device
    out = _empty(_lib.DG_OUT_COUNT, torch.float32, dev)
    perms = _empty((int(desc.n_neg), int(desc.B)), torch.long, dev)
    seed = 0
    if state is None:
        seed = int(torch.randint(0, 2 ** 62, (), dtype=torch.int64).item())
    elif state.dtype != torch.int64 or state.numel() != 3 or state.device != dev:
        raise ValueError("corr_forward_draw: state must be the int64[3] tensor of new_perm_state on the same device")
    rc = lib.dg_corr_forward_draw(ctypes.byref(desc), _ptr(feats), _ptr(feats_pos), _ptr(code), _ptr(code_pos), _ptr(depth),
                                  _ptr(coords1), _ptr(coords2), _ptr(perms), seed, _ptr(state), _ptr(out), _ptr(workspace),
                                  workspace.numel(), _stream(dev))
    _lib.check(rc, "dg_corr_forward_draw")
    return out, perms


class DeferredDropout:
    """Feature maps whose Dropout2d (`feats = self.dropout(image_feat)`, src/modules.py:122-137) has been DRAWN but not applied:
    `feats` (B,C,h,w) un-dropped, `keep` (B,C) flags 1 / 0, `scale` = 1/(1-p).  ContrastiveCorrelationLoss takes one in place of
    orig_feats / orig_feats_pos and applies the mask inside its operand preparation on the identity grid (dg_corr_forward_masked:
    the same bits, without the dropped tensor's round trip through HBM); anything else calls materialize()."""

    def __init__(self, feats, keep, scale):
        if keep.dim() != 2 or tuple(keep.shape) != tuple(feats.shape[:2]) or keep.device != feats.device:
            raise ValueError(f"depthg_amd: keep flags {tuple(keep.shape)} on {keep.device} do not match feature maps "
                             f"{tuple(feats.shape)} on {feats.device}")
        if not scale > 0:
            raise ValueError(f"depthg_amd: keep scale must be positive, got {scale}")
        self.feats, self.keep, self.scale = feats, keep, float(scale)

    shape = property(lambda self: self.feats.shape)
    device = property(lambda self: self.feats.device)
    dtype = property(lambda self: self.feats.dtype)
    is_cuda = property(lambda self: self.feats.is_cuda)

    def dim(self):
        return self.feats.dim()

    def materialize(self):
        return self.feats * (self.keep.to(torch.float32) * self.scale)[:, :, None, None]


def corr_forward_masked(desc, feats, feats_pos, code, code_pos, depth, coords1, coords2, perms, workspace, state, keep, keep_pos,
                        scale):
    """dg_corr_forward_masked: the forward on UN-dropped feature maps + their Dropout2d keep flags (identity grid).  perms None:
    drawn inside (as corr_forward_draw).  Returns (out, perms)."""
    lib = _lib.load()
    dev = feats.device
    out = _empty(_lib.DG_OUT_COUNT, torch.float32, dev)
    draw, seed = 0, 0
    if perms is None:
        draw = 1
        perms = _empty((int(desc.n_neg), int(desc.B)), torch.long, dev)
        if state is None:
            seed = int(torch.randint(0, 2 ** 62, (), dtype=torch.int64).item())
        elif state.dtype != torch.int64 or state.numel() != 3 or state.device != dev:
            raise ValueError("corr_forward_masked: state must be the int64[3] tensor of new_perm_state on the same device")
    for k in (keep, keep_pos):
        if k is not None and (k.dtype != torch.float32 or not k.is_contiguous() or k.device != dev or tuple(k.shape) != (int(desc.B), int(desc.C))):
            raise ValueError(f"corr_forward_masked: keep flags must be contiguous fp32 (B,C) = ({desc.B},{desc.C}) on {dev}")
    rc = lib.dg_corr_forward_masked(ctypes.byref(desc), _ptr(feats), _ptr(feats_pos), _ptr(code), _ptr(code_pos), _ptr(depth),
                                    _ptr(coords1), _ptr(coords2), _ptr(perms), draw, seed, _ptr(state) if draw else None,
                                    _ptr(keep), _ptr(keep_pos), float(scale), _ptr(out), _ptr(workspace), workspace.numel(),
                                    _stream(dev))
    _lib.check(rc, "dg_corr_forward_masked")
    return out, perms


def corr_backward(desc, grad_scalars, coords1, coords2, perms, workspace, shape_code):
    lib = _lib.load()
    dev = grad_scalars.device
    g_code = _empty(shape_code, torch.float32, dev)
    g_code_pos = _empty(shape_code, torch.float32, dev)
    rc = lib.dg_corr_backward(ctypes.byref(desc), _ptr(grad_scalars), _ptr(coords1), _ptr(coords2), _ptr(perms),
                              _ptr(g_code), _ptr(g_code_pos), _ptr(workspace), workspace.numel(), _stream(dev))
    _lib.check(rc, "dg_corr_backward")
    return g_code, g_code_pos


def corr_backward_total(desc, grad_total, coords1, coords2, perms, workspace, shape_code):
    """Backward for an upstream gradient of out[DG_OUT_TOTAL] alone (`total.backward()`): grad_total is a device scalar."""
    lib = _lib.load()
    dev = grad_total.device
    g_code = _empty(shape_code, torch.float32, dev)
    g_code_pos = _empty(shape_code, torch.float32, dev)
    rc = lib.dg_corr_backward_total(ctypes.byref(desc), _ptr(grad_total), _ptr(coords1), _ptr(coords2), _ptr(perms),
                                    _ptr(g_code), _ptr(g_code_pos), _ptr(workspace), workspace.numel(), _stream(dev))
    _lib.check(rc, "dg_corr_backward_total")
    return g_code, g_code_pos


def corr_materialize(desc, which, workspace, want_cd=True, want_loss=False, perms=None):
    """The un-reduced (B,S,S,S,S) tensors of pair-set `which` (-1: the depth term's dd) from the operands the forward left in
    `workspace`.  `perms`: the forward's (n_neg, B) batch maps - needed for the negatives of a shared-coordinates (dense grid) call."""
    lib = _lib.load()
    dev = workspace.device
    sh = 1 if (desc.flags & _lib.DG_LINE_GRID) else desc.S
    shape = (desc.B, sh, desc.S, sh, desc.S)
    cd = _empty(shape, torch.float32, dev) if want_cd else None
    loss = _empty(shape, torch.float32, dev) if want_loss else None
    rc = lib.dg_corr_materialize_shared(ctypes.byref(desc), int(which), _ptr(perms), _ptr(cd), _ptr(loss), _ptr(workspace),
                                        workspace.numel(), _stream(dev))
    _lib.check(rc, "dg_corr_materialize_shared")
    return cd, loss


def fps_coords(depth, feat_hw, n_samples, return_inds=False):
    """depth (B,1,H,W) on GPU -> coords (B,S,S,2) in [-1,1) (already *2-1)."""
    lib = _lib.load()
    depth = _f32c(depth, "depth")
    B, _, H, W = depth.shape
    h, w = int(feat_hw[0]), int(feat_hw[1])
    S = int(n_samples)
    coords = _empty((B, S, S, 2), torch.float32, depth.device)
    inds = _empty((B, S * S), torch.int32, depth.device) if return_inds else None
    ws = _empty((lib.dg_fps_workspace_bytes(B, h, w)), torch.uint8, depth.device)
    rc = lib.dg_fps_coords(_ptr(depth), B, H, W, h, w, S, _ptr(coords), _ptr(inds), _ptr(ws), ws.numel(),
                           _stream(depth.device))
    _lib.check(rc, "dg_fps_coords")
    return (coords, inds) if return_inds else coords


def fps_coords_pair(depth, depth_pos, feat_hw, n_samples):
    """The two FPS calls of one step in one launch (dg_fps_coords_pair): depth, depth_pos (B,1,H,W) -> coords (2B,S,S,2), rows
    [0,B) the anchors', [B,2B) the positives'."""
    lib = _lib.load()
    depth, depth_pos = _f32c(depth, "depth"), _f32c(depth_pos, "depth_pos")
    if depth.shape != depth_pos.shape or depth.device != depth_pos.device:
        raise ValueError(f"depthg_amd: depth {tuple(depth.shape)} on {depth.device} and depth_pos {tuple(depth_pos.shape)} on "
                         f"{depth_pos.device} must match for the paired sampler")
    B, _, H, W = depth.shape
    h, w = int(feat_hw[0]), int(feat_hw[1])
    S = int(n_samples)
    coords = _empty((2 * B, S, S, 2), torch.float32, depth.device)
    ws = _empty((lib.dg_fps_workspace_bytes(2 * B, h, w)), torch.uint8, depth.device)      # the pooled depth maps of both calls
    rc = lib.dg_fps_coords_pair(_ptr(depth), _ptr(depth_pos), B, H, W, h, w, S, _ptr(coords), None, _ptr(ws), ws.numel(), _stream(depth.device))
    _lib.check(rc, "dg_fps_coords_pair")
    return coords


def salience_coords(salience, n_side, u_sel=None, u_fallback=None):
    """sample_nonzero_locations (src/modules.py:1191-1204) on the GPU: salience (B,H,W) -> (B,S,S,2), flipped and *2-1 like
    the reference.  u_sel (B,S*S) / u_fallback (B,S*S,2): iid uniforms in [0,1), drawn here unless given (tests)."""
    lib = _lib.load()
    sal = _f32c(salience, "salience")
    if sal.dim() != 3:
        raise ValueError(f"depthg_amd: salience must be (B,H,W) like the reference's batch['mask'].squeeze(1), got {tuple(sal.shape)}")
    B, H, W = sal.shape
    S = int(n_side)
    n = S * S
    dev = sal.device
    u_sel = torch.rand(B, n, device=dev) if u_sel is None else _f32c(u_sel, "u_sel")
    u_fallback = torch.rand(B, n, 2, device=dev) if u_fallback is None else _f32c(u_fallback, "u_fallback")
    assert tuple(u_sel.shape) == (B, n) and tuple(u_fallback.shape) == (B, n, 2)
    coords = _empty((B, S, S, 2), torch.float32, dev)
    rc = lib.dg_salience_coords(_ptr(sal), B, H, W, n, _ptr(u_sel), _ptr(u_fallback), _ptr(coords), _stream(dev))
    _lib.check(rc, "dg_salience_coords")
    return coords


def simple_depth_coords(depth, feat_hw, n_samples, u_value=None, u_pick=None):
    """simple_depth_informed_sampling (src/modules.py:828-883) on the GPU: depth (B,1,H,W) -> coords (B,n,1,2), already
    *2-1 (the caller's step at modules.py:1300).  u_value / u_pick (B,n): iid uniforms, drawn here unless given."""
    lib = _lib.load()
    depth = _f32c(depth, "depth")
    B, _, H, W = depth.shape
    h, w = int(feat_hw[0]), int(feat_hw[1])
    n = int(n_samples)
    dev = depth.device
    u_value = torch.rand(B, n, device=dev) if u_value is None else _f32c(u_value, "u_value")
    u_pick = torch.rand(B, n, device=dev) if u_pick is None else _f32c(u_pick, "u_pick")
    assert tuple(u_value.shape) == (B, n) and tuple(u_pick.shape) == (B, n)
    coords = _empty((B, n, 1, 2), torch.float32, dev)
    rc = lib.dg_simple_depth_coords(_ptr(depth), B, H, W, h, w, n, _ptr(u_value), _ptr(u_pick), _ptr(coords), _stream(dev))
    _lib.check(rc, "dg_simple_depth_coords")
    return coords


def confusion_update(stats, preds, target, n_classes, extra_clusters):
    """stats (n_classes + extra, n_classes) int64 on the GPU += confusion counts of (preds, target) (src/utils.py:222-232)."""
    lib = _lib.load()
    for name, t in (("stats", stats), ("preds", preds), ("target", target)):
        if not t.is_cuda:
            raise RuntimeError(f"depthg_amd: `{name}` must live on the GPU (got {t.device}); there is no CPU path")
    if stats.dtype != torch.int64 or not stats.is_contiguous() or tuple(stats.shape) != (n_classes + extra_clusters, n_classes):
        raise ValueError("depthg_amd: stats must be a contiguous int64 (n_classes + extra_clusters, n_classes) tensor")
    p = preds.detach().reshape(-1).to(torch.int64).contiguous()
    a = target.detach().reshape(-1).to(torch.int64).contiguous()
    if p.numel() != a.numel():
        raise ValueError(f"depthg_amd: preds and target differ in size ({p.numel()} vs {a.numel()})")
    rc = lib.dg_confusion_update(_ptr(p), _ptr(a), p.numel(), int(n_classes), int(extra_clusters), _ptr(stats), _stream(stats.device))
    _lib.check(rc, "dg_confusion_update")
    return stats


def topk_rows(vals, k, return_values=False):
    """Column indices of the k largest entries of every row of `vals` (rows, cols) fp32 on the GPU: value descending, ties by
    ascending column (src/precompute_knns.py:110 `torch.topk(pairwise_sims, 30)[1]`)."""
    lib = _lib.load()
    if not vals.is_cuda:
        raise RuntimeError(f"depthg_amd: `vals` must live on the GPU (got {vals.device}); there is no CPU path")
    if vals.dim() != 2 or vals.dtype != torch.float32 or vals.stride(1) != 1:
        raise ValueError("depthg_amd: topk_rows wants a 2-D fp32 tensor with contiguous rows")
    rows, cols = vals.shape
    idx = _empty((rows, int(k)), torch.int64, vals.device)
    val = _empty((rows, int(k)), torch.float32, vals.device) if return_values else None
    rc = lib.dg_topk_rows(_ptr(vals), rows, cols, vals.stride(0) if rows > 1 else cols, int(k), _ptr(idx), _ptr(val), _stream(vals.device))
    _lib.check(rc, "dg_topk_rows")
    return (idx, val) if return_values else idx


def knn_similarities(queries, feats):
    """queries (m, F), feats (n, F) fp32 on the GPU (rows contiguous) -> (m, n) fp32 similarities, `einsum("nf,mf->nm")` of
    src/precompute_knns.py:106-108 on the fp32 MFMA."""
    lib = _lib.load()
    for name, t in (("queries", queries), ("feats", feats)):
        if not t.is_cuda:
            raise RuntimeError(f"depthg_amd: `{name}` must live on the GPU (got {t.device}); there is no CPU path")
        if t.dim() != 2 or t.dtype != torch.float32 or t.stride(1) != 1:
            raise ValueError(f"depthg_amd: `{name}` must be a 2-D fp32 tensor with contiguous rows")
    if queries.shape[1] != feats.shape[1] or queries.device != feats.device:
        raise RuntimeError(f"depthg_amd: queries {tuple(queries.shape)} on {queries.device} and feats {tuple(feats.shape)} on {feats.device} do not match")
    m, F = queries.shape
    n = feats.shape[0]
    out = _empty((m, n), torch.float32, feats.device)
    rc = lib.dg_knn_similarities(_ptr(queries), _ptr(feats), m, n, F, queries.stride(0) if m > 1 else F, feats.stride(0) if n > 1 else F,
                                 _ptr(out), n, _stream(feats.device))
    _lib.check(rc, "dg_knn_similarities")
    return out


def lhp_forward(code, depth):
    """code (B,D,h,w), depth (B,1,H,W) on the GPU -> (code_mixed, points, stats) of dg_lhp_forward."""
    lib = _lib.load()
    code = _f32c(code, "code")
    depth = _f32c(depth, "depth")
    B, D, h, w = code.shape
    out = _empty(tuple(code.shape), torch.float32, code.device)
    points = _empty((B, 3, h * w), torch.float32, code.device)
    stats = _empty((B, h * w, 3), torch.float32, code.device)
    rc = lib.dg_lhp_forward(_ptr(code), _ptr(depth), B, D, h, w, depth.shape[-2], depth.shape[-1], _ptr(out), _ptr(points),
                            _ptr(stats), _stream(code.device))
    _lib.check(rc, "dg_lhp_forward")
    return out, points, stats


def lhp_backward(grad_out, points, stats):
    lib = _lib.load()
    g = _f32c(grad_out, "grad_out")
    B, D, h, w = g.shape
    grad_code = _empty(tuple(g.shape), torch.float32, g.device)
    rc = lib.dg_lhp_backward(_ptr(g), _ptr(points), _ptr(stats), B, D, h, w, _ptr(grad_code), _stream(g.device))
    _lib.check(rc, "dg_lhp_backward")
    return grad_code


LHP_ATTN, LHP_ORIG_DEPTH, LHP_ORIG_ATTN = 0, 1, 2          # enum of include/depthg_corr.h


def lhp_map_forward(mode, code, attn=None, depth=None, divide=None):
    """dg_lhp_map_forward: code (B,D,h,w) with attn (B,heads,h*w+1,h*w+1) or depth (B,1,H,W) -> (code_mixed, map).
    `map` is what dg_lhp_map_backward needs: (B,P,P) for LHP_ATTN, (B,P,9) for the Original variants."""
    lib = _lib.load()
    code = _f32c(code, "code")
    B, D, h, w = code.shape
    P = h * w
    heads, dh, dw, points = 0, 0, 0, None
    if mode == LHP_ORIG_DEPTH:
        depth = _f32c(depth, "depth")
        dh, dw = depth.shape[-2], depth.shape[-1]
        points = _empty((B, 3, P), torch.float32, code.device)
    else:
        attn = _f32c(attn, "attn")
        if attn.dim() != 4 or attn.shape[0] != B or attn.shape[2] != P + 1 or attn.shape[3] != P + 1:
            raise ValueError(f"attn must be (B, heads, {P + 1}, {P + 1}) for a {h}x{w} code map, got {tuple(attn.shape)}")
        heads = attn.shape[1]
    if mode != LHP_ATTN:
        divide = _f32c(divide, "divide")
        if divide.numel() != P:
            raise ValueError(f"divide must hold {P} divisors, got {divide.numel()}")
    out = _empty(tuple(code.shape), torch.float32, code.device)
    wmap = _empty((B, P, P if mode == LHP_ATTN else 9), torch.float32, code.device)
    rc = lib.dg_lhp_map_forward(mode, _ptr(code), _ptr(attn) if mode != LHP_ORIG_DEPTH else None,
                                _ptr(depth) if mode == LHP_ORIG_DEPTH else None, _ptr(divide) if mode != LHP_ATTN else None,
                                B, D, h, w, heads, dh, dw, _ptr(out), _ptr(wmap), _ptr(points), _stream(code.device))
    _lib.check(rc, "dg_lhp_map_forward")
    return out, wmap


def lhp_map_backward(mode, grad_out, wmap, divide=None):
    lib = _lib.load()
    g = _f32c(grad_out, "grad_out")
    B, D, h, w = g.shape
    grad_code = _empty(tuple(g.shape), torch.float32, g.device)
    rc = lib.dg_lhp_map_backward(mode, _ptr(g), _ptr(wmap), _ptr(divide) if mode != LHP_ATTN else None, B, D, h, w,
                                 _ptr(grad_code), _stream(g.device))
    _lib.check(rc, "dg_lhp_map_backward")
    return grad_code


def new_perm_state(device):
    """Device-resident generator state for super_perms(state=...): int64 {seed, draws so far, 0}; the seed comes from torch's
    CPU generator, so torch.manual_seed before the first use fixes the whole sequence."""
    seed = int(torch.randint(0, 2 ** 62, (), dtype=torch.int64).item())
    return torch.tensor([seed, 0, 0], dtype=torch.int64, device=device)


def rand_coords_state(state, shape):
    """Two coordinate tensors of `shape` (B, S, S2, 2), uniform in [-1, 1), from the device-resident generator `state`
    (new_perm_state) in ONE launch - for steps recorded in a hipGraph (dg_rand_coords_state); advances the state."""
    lib = _lib.load()
    if state.dtype != torch.int64 or state.numel() != 3 or not state.is_cuda:
        raise ValueError("rand_coords_state: state must be the int64[3] device tensor of new_perm_state")
    both = _empty((2,) + tuple(int(v) for v in shape), torch.float32, state.device)
    rc = lib.dg_rand_coords_state(_ptr(state), both.numel(), _ptr(both), _stream(state.device))
    _lib.check(rc, "dg_rand_coords_state")
    return both[0], both[1]


def keep_masks_state(state, rows, C, p=0.1, use=(True, True, True)):
    """The Dropout2d keep masks of a graph-recorded step from the device-resident generator, ONE launch: a tuple of three (rows, C)
    tensors of 1 / 0 (keep with probability 1 - p), None where `use` is False - what ProjectionHead.forward(_pair) takes as `keeps`
    (rows = B, or 2B for forward_pair).  Advances the state; not torch's random stream."""
    lib = _lib.load()
    if state.dtype != torch.int64 or state.numel() != 3 or not state.is_cuda:
        raise ValueError("keep_masks_state: state must be the int64[3] device tensor of new_perm_state")
    k = sum(1 for u in use if u)
    if k == 0:
        return (None, None, None)
    buf = _empty((k, int(rows), int(C)), torch.float32, state.device)
    rc = lib.dg_rand_keep_state(_ptr(state), buf.numel(), float(1.0 - p), _ptr(buf), _stream(state.device))
    _lib.check(rc, "dg_rand_keep_state")
    it = iter(range(k))
    return tuple(buf[next(it)] if u else None for u in use)


def super_perms(count, size, device, keys=None, state=None):
    """(count, size) int64: independent super_perm draws (src/modules.py:1184-1188), one kernel.  `state` (new_perm_state):
    the draw is keyed by device memory and advances it - safe to record in a hipGraph (every replay draws anew)."""
    lib = _lib.load()
    out = _empty((count, size), torch.long, device)
    if count == 0:
        return out
    if state is not None:
        if state.dtype != torch.int64 or state.numel() != 3 or state.device != out.device:
            raise ValueError("super_perms: state must be the int64[3] tensor of new_perm_state on the same device")
        rc = lib.dg_super_perms_state(_ptr(state), int(count), int(size), _ptr(out), _stream(out.device))
        _lib.check(rc, "dg_super_perms_state")
        return out
    if keys is None:
        # one launch: the keys are drawn inside the kernel (Philox) from a 64-bit seed taken from torch's CPU generator, so
        # torch.manual_seed still fixes the sequence and no device RNG launch is needed
        seed = int(torch.randint(0, 2 ** 62, (), dtype=torch.int64).item())
        rc = lib.dg_super_perms_seeded(seed, int(count), int(size), _ptr(out), _stream(out.device))
        _lib.check(rc, "dg_super_perms_seeded")
        return out
    rc = lib.dg_super_perms(_ptr(keys), int(count), int(size), _ptr(out), _stream(out.device))
    _lib.check(rc, "dg_super_perms")
    return out


def corr_relaunch_main(desc, perms, workspace):
    """Measurement aid: launch only the fused correlation kernel again (operands already in `workspace`)."""
    lib = _lib.load()
    rc = lib.dg_corr_relaunch_main(ctypes.byref(desc), _ptr(perms), _ptr(workspace), workspace.numel(),
                                   _stream(workspace.device))
    _lib.check(rc, "dg_corr_relaunch_main")


class MainKernelTimer:
    """Measurement aid (bench.py): the execution span of the fused correlation launch INSIDE the step, hipGraph replays included
    (dg_prof_main_span): while armed, every workgroup of that kernel stamps its entry / exit time (the GPU's constant 100-MHz
    clock) into two device words with atomic min / max and adds its lifetime in shader cycles / wall ticks to two more.
    `reset()` in front of a step (an asynchronous copy on the current stream), `last_ms()` / `last()` behind it: the interval a
    kernel trace reports for the launch, minus the dispatch ramp, and the clock the kernel's CUs held.  A hipGraph captured while
    the timer is armed keeps writing to `span` at every replay: keep the timer alive as long as such a graph."""

    def __init__(self, device):
        self.span = torch.zeros(4, dtype=torch.int64, device=device)
        self._init = torch.tensor([-1, 0, 0, 0], dtype=torch.int64, device=device)       # {UINT64_MAX, 0, 0, 0}

    def arm(self, on=True):
        _lib.check(_lib.load().dg_prof_main_span(_ptr(self.span) if on else None), "dg_prof_main_span")

    def reset(self):
        self.span.copy_(self._init, non_blocking=True)

    def last(self):
        """(span in ms, held shader clock in GHz) of the launches since the last reset; (nan, nan) when nothing was stamped."""
        torch.cuda.synchronize()
        t0, t1, cyc, ticks = (int(v) for v in self.span.tolist())
        if t0 < 0 or t1 <= 0:                      # nothing stamped
            return float("nan"), float("nan")
        ghz = cyc / ticks * 0.1 if ticks > 0 else float("nan")      # cycles per 10-ns tick
        return (t1 - t0) * 1e-5, ghz               # 10-ns ticks -> ms

    def last_ms(self):
        return self.last()[0]


def corr_intra_folded(desc):
    """True when the fused correlation launch of `desc` also forms the intra pair-set's streamed-side gradient (k_corr2's FOLD)."""
    rc = _lib.load().dg_corr_intra_folded(ctypes.byref(desc))
    if rc < 0:
        _lib.check(-1, "dg_corr_intra_folded")
    return rc == 1


def corr_main_kernel_name(desc):
    """Which kernel the fused correlation launch of `desc` runs ("k_corr2" / "k_corr_main"), from the library's own predicate."""
    name = _lib.load().dg_corr_main_kernel_name(ctypes.byref(desc))
    if name is None:
        _lib.check(-1, "dg_corr_main_kernel_name")
    return name.decode()
