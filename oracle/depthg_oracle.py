"""CPU oracle for the DepthG correlation-loss hot path.  TEST INFRASTRUCTURE ONLY.

This file is a restatement, written from the maths in SURVEY.md section 9, of what the
reference computes on the path `ContrastiveCorrelationLoss.forward` and its helpers.  It is
the *checker* used by `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py`.  Nothing in the product package (`depthg_amd/`) may import it.

Pinning: the reference ships no tests, known-answer vectors or fixtures for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference itself,
imported on CPU in the build container by `tests/golden/make_fixtures.py` and stored as
`tests/golden/*.npz`.  `tests/test_oracle_golden.py` checks every function below against them.

Arithmetic is float32 on torch-CPU / numpy, like the reference.  The third-party pieces the
reference leans on (`F.normalize`, `einsum`, `F.grid_sample`, `F.interpolate`,
`F.adaptive_avg_pool2d`, numpy argmax/minimum) are restated from their published definitions
instead of being called, so that the HIP kernels have an explicit formula to follow.

Every function cites the reference lines it follows (paths relative to the reference root).
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

EPS_NORM = 1e-10  # src/modules.py:790


# --------------------------------------------------------------------------------------
# A1  norm                                                             src/modules.py:789-790
# --------------------------------------------------------------------------------------
def norm(t: torch.Tensor) -> torch.Tensor:
    """L2-normalise over dim 1: t / max(||t||_2, 1e-10)."""
    n = t.square().sum(dim=1, keepdim=True).sqrt()
    return t / n.clamp_min(EPS_NORM)


# --------------------------------------------------------------------------------------
# A2/A3  tensor_correlation / depth_correlation                        src/modules.py:797-814
# --------------------------------------------------------------------------------------
def tensor_correlation(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """out[n,h,w,i,j] = sum_c a[n,c,h,w] * b[n,c,i,j]  (a batched (P x C)(C x P) product)."""
    n, c, h, w = a.shape
    _, _, i, j = b.shape
    am = a.reshape(n, c, h * w).transpose(1, 2)  # (n, P1, c)
    bm = b.reshape(n, c, i * j)                  # (n, c, P2)
    return torch.bmm(am, bm).reshape(n, h, w, i, j)


depth_correlation = tensor_correlation  # same contraction with c == 1 (src/modules.py:812-814)


# --------------------------------------------------------------------------------------
# A6  sample = grid_sample(t, coords.permute(0,2,1,3), bilinear, border, align_corners=True)
#                                                                      src/modules.py:822-825
# --------------------------------------------------------------------------------------
def bilinear_taps(coords: torch.Tensor, h: int, w: int):
    """Tap positions/weights of `sample` for coords (B,S,S,2) in [-1,1].

    Returns (x0, y0, x1, y1, wx1, wy1) each shaped (B,S,S) *in output order* (i, j), i.e.
    already including the (0,2,1,3) permute: output position (i, j) reads
    x = coords[b, j, i, 0] (width axis), y = coords[b, j, i, 1] (height axis)  (quirk Q3).
    Unnormalise with align_corners=True: ((c + 1) / 2) * (size - 1); clip to [0, size-1]
    (padding_mode='border'); taps floor / floor+1, the +1 tap is dropped when outside.
    """
    g = coords.permute(0, 2, 1, 3).to(torch.float32)
    x = ((g[..., 0] + 1.0) / 2.0) * float(w - 1)
    y = ((g[..., 1] + 1.0) / 2.0) * float(h - 1)
    x = x.clamp(0.0, float(w - 1))
    y = y.clamp(0.0, float(h - 1))
    x0 = x.floor()
    y0 = y.floor()
    wx1 = x - x0
    wy1 = y - y0
    x0 = x0.long()
    y0 = y0.long()
    return x0, y0, x0 + 1, y0 + 1, wx1, wy1


def sample(t: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """t (B,K,h,w), coords (B,S,S,2) -> (B,K,S,S)."""
    b, k, h, w = t.shape
    x0, y0, x1, y1, wx1, wy1 = bilinear_taps(coords, h, w)
    wx0 = 1.0 - wx1
    wy0 = 1.0 - wy1
    s1, s2 = x0.shape[1], x0.shape[2]
    flat = t.reshape(b, k, h * w)

    def tap(yy, xx, wgt):
        inside = ((xx <= w - 1) & (yy <= h - 1)).to(t.dtype)
        idx = (yy.clamp(max=h - 1) * w + xx.clamp(max=w - 1)).reshape(b, 1, s1 * s2).expand(b, k, s1 * s2)
        v = torch.gather(flat, 2, idx).reshape(b, k, s1, s2)
        return v * (wgt * inside).unsqueeze(1)

    out = tap(y0, x0, wy0 * wx0) + tap(y0, x1, wy0 * wx1) + tap(y1, x0, wy1 * wx0) + tap(y1, x1, wy1 * wx1)
    return out


# --------------------------------------------------------------------------------------
# F.interpolate(d, size, mode='bilinear', align_corners=True)          src/modules.py:1261-1262
# --------------------------------------------------------------------------------------
def interpolate_bilinear_ac(d: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    b, c, hin, win = d.shape
    hout, wout = size

    def axis(nin, nout):
        scale = np.float32(nin - 1) / np.float32(nout - 1) if nout > 1 else np.float32(0.0)
        src = torch.arange(nout, dtype=torch.float32) * float(scale)
        i0 = src.floor().long().clamp(max=nin - 1)
        i1 = torch.where(i0 < nin - 1, i0 + 1, i0)
        l1 = src - i0.to(torch.float32)
        return i0, i1, 1.0 - l1, l1

    y0, y1, ly0, ly1 = axis(hin, hout)
    x0, x1, lx0, lx1 = axis(win, wout)
    top = d[:, :, y0][:, :, :, x0] * lx0 + d[:, :, y0][:, :, :, x1] * lx1
    bot = d[:, :, y1][:, :, :, x0] * lx0 + d[:, :, y1][:, :, :, x1] * lx1
    return top * ly0.view(1, 1, -1, 1) + bot * ly1.view(1, 1, -1, 1)


# --------------------------------------------------------------------------------------
# F.adaptive_avg_pool2d(depth, (h, w))                                 src/modules.py:1003
# --------------------------------------------------------------------------------------
def adaptive_avg_pool2d(d: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """Window [floor(i*H/h), ceil((i+1)*H/h)) x the same along x; the window is summed ROW-MAJOR AND SEQUENTIALLY in
    float32, then divided by the window height and by the window width (two divisions): the order of the torch CPU operator the reference calls (bit-exact against
    F.adaptive_avg_pool2d, pinned in tests/test_oracle_golden.py; a pairwise/vectorised sum differs in the last bit on
    general float depth, enough to swap near-tied FPS picks)."""
    x = d.detach().cpu().numpy().astype(np.float32, copy=False)
    b, c, hin, win = x.shape
    hout, wout = size
    out = np.empty((b, c, hout, wout), dtype=np.float32)
    if hin % hout == 0 and win % wout == 0:                   # equal windows: all outputs at once, one add per window element
        kh, kw = hin // hout, win // wout
        v = x.reshape(b, c, hout, kh, wout, kw)
        acc = np.zeros((b, c, hout, wout), dtype=np.float32)
        for y in range(kh):
            for xx in range(kw):
                acc = acc + v[:, :, :, y, :, xx]
        out = acc / np.float32(kh) / np.float32(kw)
    else:
        for i in range(hout):
            ys, ye = (i * hin) // hout, -((-(i + 1) * hin) // hout)
            for j in range(wout):
                xs, xe = (j * win) // wout, -((-(j + 1) * win) // wout)
                acc = np.zeros((b, c), dtype=np.float32)
                for y in range(ys, ye):
                    for xx in range(xs, xe):
                        acc = acc + x[:, :, y, xx]
                out[:, :, i, j] = acc / np.float32(ye - ys) / np.float32(xe - xs)
    return torch.from_numpy(np.ascontiguousarray(out)).to(d.dtype)


# --------------------------------------------------------------------------------------
# A9  depth2points                                                     src/modules.py:988-996
# --------------------------------------------------------------------------------------
def fov_factor(fov: float = 90.0) -> torch.Tensor:
    """2*tan(fov/2) with fov taken in RADIANS (quirk Q5), float32."""
    return 2.0 * torch.tan(torch.tensor([fov], dtype=torch.float32) / 2.0)


def depth2points(depth: torch.Tensor, fov: float = 30.0, far: float = 5.0) -> torch.Tensor:
    """depth (h,w) -> (3,h,w) = [X, Y, Z];  X,Y = factor*d*(idx - n/2)/n, Z = -d*far."""
    h, w = depth.shape[-2], depth.shape[-1]
    factor = fov_factor(fov).to(depth.dtype)
    rows = torch.arange(h).view(h, 1).expand(h, w)
    cols = torch.arange(w).view(1, w).expand(h, w)
    y = factor * depth * (rows - h / 2.0) / h
    x = factor * depth * (cols - w / 2.0) / w
    return torch.stack([x, y, -depth * far])


# --------------------------------------------------------------------------------------
# A10  fps                                                             src/modules.py:939-985
# --------------------------------------------------------------------------------------
def fps(points, n_samples: int) -> np.ndarray:
    """Farthest point sampling, start at index 0, squared-L2 in float32, first-max ties.

    Returns the `n_samples` selected indices in selection order.  Restated without
    `np.delete`: a boolean `left` mask plays the role of `points_left`; since the reference's
    `points_left` stays in ascending order, "first max over points_left" == "lowest original
    index among the maxima of the unselected points".
    """
    pts = np.asarray(points, dtype=np.float32)
    n = pts.shape[0]
    left = np.ones(n, dtype=bool)
    dists = np.full(n, np.inf, dtype=np.float64)
    out = np.zeros(n_samples, dtype=np.int64)
    out[0] = 0
    left[0] = False
    for i in range(1, n_samples):
        last = pts[out[i - 1]]
        diff = last[None, :] - pts
        sq = diff * diff
        d = (sq[:, 0] + sq[:, 1]) + sq[:, 2]          # float32, x,y,z order
        dists = np.where(left, np.minimum(d.astype(np.float64), dists), dists)
        masked = np.where(left, dists, -np.inf)
        sel = int(np.argmax(masked))
        out[i] = sel
        left[sel] = False
    return out


# --------------------------------------------------------------------------------------
# A11  farthest_point_sampling_depth                                   src/modules.py:999-1037
# --------------------------------------------------------------------------------------
def farthest_point_sampling_depth(feat_hw: Tuple[int, int], depth: torch.Tensor, n_samples: int,
                                  return_inds: bool = False):
    """depth (B,1,H,W) -> coords (B,S,S,2) in [0,(h-1)/h]: row/h in [...,0], col/w in [...,1].

    The selected *set* is re-emitted in row-major order (quirk Q4).  The caller maps *2-1.
    """
    h, w = feat_hw
    d = adaptive_avg_pool2d(depth, (h, w))
    coords_all = []
    inds_all = []
    for i in range(d.shape[0]):
        pc = depth2points(d[i, 0], fov=90.0).permute(1, 2, 0).reshape(-1, 3)
        inds = fps(pc.numpy(), n_samples ** 2)
        inds_all.append(inds)
        srt = np.sort(inds)
        rows = torch.from_numpy((srt // w).astype(np.float32)) / h
        cols = torch.from_numpy((srt % w).astype(np.float32)) / w
        coords_all.append(torch.stack([rows, cols], dim=-1).reshape(n_samples, n_samples, 2))
    coords = torch.stack(coords_all, dim=0)
    if return_inds:
        return coords, np.stack(inds_all)
    return coords


# --------------------------------------------------------------------------------------
# A8  super_perm                                                       src/modules.py:1184-1188
# --------------------------------------------------------------------------------------
def super_perm(size: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """randperm with fixed points bumped by one, modulo size (may contain duplicates, Q6)."""
    perm = torch.randperm(size, generator=generator, dtype=torch.long)
    return super_perm_from(perm)


def super_perm_from(perm: torch.Tensor) -> torch.Tensor:
    size = perm.numel()
    bumped = torch.where(perm == torch.arange(size), perm + 1, perm)
    return bumped % size


# --------------------------------------------------------------------------------------
# N4  the other two samplers of the coordinate draw
# --------------------------------------------------------------------------------------
def rank_from_uniform(u, count: int) -> int:
    """Rank among `count` candidates from a uniform u in [0,1): float32 product, truncated, clamped.  This is the
    build's device-side replacement for the reference's host-side torch.randint(count) (include/depthg_corr.h)."""
    r = int(np.float32(u) * np.float32(count))
    return min(max(r, 0), count - 1)


def nonzero_by_ranks(t_img: torch.Tensor, ranks: np.ndarray) -> np.ndarray:
    """(n,) ranks -> (n,2) int64 (row, col) of the rank-th non-zero of a (H,W) map in row-major (torch.nonzero) order."""
    h, w = t_img.shape
    flat = np.flatnonzero(t_img.reshape(-1).numpy() != 0)
    sel = flat[np.asarray(ranks, dtype=np.int64)]
    return np.stack([sel // w, sel % w], axis=-1)


def _salience_finish(picked: np.ndarray, target_size, h: int) -> torch.Tensor:
    """(B,n,2) integer (row, col) -> what src/modules.py:1201-1204 returns: / t.shape[1] (BOTH coordinates), *2-1, flipped."""
    c = torch.from_numpy(picked.reshape(target_size).astype(np.int64)).to(torch.float32) / h
    c = c * 2 - 1
    return torch.flip(c, dims=[-1])


def sample_nonzero_locations(t: torch.Tensor, target_size) -> torch.Tensor:
    """src/modules.py:1191-1204 with the reference's RNG calls in the reference's order (one torch.randint per image:
    ranks among the image's non-zeros, or (n,2) integers in [0,H) for an image without any)."""
    n = target_size[1] * target_size[2]
    picked = np.zeros((t.shape[0], n, 2), dtype=np.int64)
    for i in range(t.shape[0]):
        count = int((t[i] != 0).sum())
        if count == 0:
            picked[i] = torch.randint(t.shape[1], size=(n, 2)).numpy()
        else:
            picked[i] = nonzero_by_ranks(t[i], torch.randint(count, size=(n,)).numpy())
    return _salience_finish(picked, target_size, t.shape[1])


def sample_nonzero_locations_from_uniform(t: torch.Tensor, target_size, u_sel, u_fallback) -> torch.Tensor:
    """Same selection with caller-provided uniforms (dg_salience_coords): u_sel (B,n), u_fallback (B,n,2)."""
    n = target_size[1] * target_size[2]
    hh = t.shape[1]
    picked = np.zeros((t.shape[0], n, 2), dtype=np.int64)
    for i in range(t.shape[0]):
        count = int((t[i] != 0).sum())
        if count == 0:
            picked[i] = [[rank_from_uniform(u_fallback[i, s, 0], hh), rank_from_uniform(u_fallback[i, s, 1], hh)] for s in range(n)]
        else:
            picked[i] = nonzero_by_ranks(t[i], [rank_from_uniform(u_sel[i, s], count) for s in range(n)])
    return _salience_finish(picked, target_size, hh)


def adaptive_max_pool2d(d: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """F.adaptive_max_pool2d (src/modules.py:830): window [floor(i*H/h), ceil((i+1)*H/h))."""
    b, c, hin, win = d.shape
    hout, wout = size
    out = torch.empty(b, c, hout, wout, dtype=d.dtype)
    for i in range(hout):
        ys, ye = (i * hin) // hout, -((-(i + 1) * hin) // hout)
        for j in range(wout):
            xs, xe = (j * win) // wout, -((-(j + 1) * win) // wout)
            out[:, :, i, j] = d[:, :, ys:ye, xs:xe].amax(dim=(2, 3))
    return out


def _rounded_depth(feat_hw, depth: torch.Tensor) -> torch.Tensor:
    d = adaptive_max_pool2d(depth, tuple(feat_hw))
    return (d * 10).round() / 10                                                   # src/modules.py:832


def _simple_finish(rows_cols: np.ndarray, feat_hw) -> torch.Tensor:
    """(B,n,2) integer (row, col) -> (B,n,1,2) = (coord + 0.5) / (h, w)   (src/modules.py:872-881)."""
    c = (torch.from_numpy(rows_cols.astype(np.int64)).to(torch.float32) + 0.5) / torch.tensor(
        [feat_hw[0], feat_hw[1]], dtype=torch.float32)
    return c.unsqueeze(2)


def simple_depth_informed_sampling(feat_hw, depth: torch.Tensor, n_samples: int) -> torch.Tensor:
    """src/modules.py:828-883 with the reference's RNG calls in the reference's order: one torch.multinomial per image
    over the unique-value frequencies (all images first), then one torch.randint per (image, sample) among the pixels
    holding the drawn value (torch.nonzero = row-major order).  Returns (B,n,1,2) in [0,1] as (row, col)."""
    d = _rounded_depth(feat_hw, depth)
    b = d.shape[0]
    uniq, drawn = [], []
    for i in range(b):
        vals, counts = np.unique(d[i].numpy(), return_counts=True)               # sorted, like torch.unique
        uniq.append(vals)
        cnt = torch.from_numpy(counts)
        probs = cnt.float() / cnt.sum()
        drawn.append(torch.multinomial(probs, n_samples, replacement=True).numpy())
    out = np.zeros((b, n_samples, 2), dtype=np.int64)
    for i in range(b):
        plane = d[i, 0]
        for s in range(n_samples):
            same = (plane == float(uniq[i][drawn[i][s]])).to(torch.float32)
            count = int(same.sum())
            out[i, s] = nonzero_by_ranks(same, [int(torch.randint(count, (1,)))])[0]
    return _simple_finish(out, feat_hw)


def simple_depth_informed_sampling_from_uniform(feat_hw, depth: torch.Tensor, n_samples: int, u_value, u_pick) -> torch.Tensor:
    """Same two-stage draw with caller-provided uniforms (dg_simple_depth_coords): the value is the one at sorted
    position R = rank_from_uniform(u_value, h*w) (probability count / (h*w), like the multinomial), the pixel the
    rank_from_uniform(u_pick, count)-th holder of that value in row-major order."""
    d = _rounded_depth(feat_hw, depth)
    b = d.shape[0]
    hw = int(feat_hw[0]) * int(feat_hw[1])
    out = np.zeros((b, n_samples, 2), dtype=np.int64)
    for i in range(b):
        plane = d[i, 0]
        srt = np.sort(plane.reshape(-1).numpy(), kind="stable")
        for s in range(n_samples):
            v = srt[rank_from_uniform(u_value[i, s], hw)]
            same = (plane == float(v)).to(torch.float32)
            out[i, s] = nonzero_by_ranks(same, [rank_from_uniform(u_pick[i, s], int(same.sum()))])[0]
    return _simple_finish(out, feat_hw)


# --------------------------------------------------------------------------------------
# N3  LocalHiddenPositiveProjection.forward_depth (before the projection head)   src/modules.py:273-335
# --------------------------------------------------------------------------------------
def quantile_lerp(sorted_row: np.ndarray, q: float) -> np.float32:
    """torch.quantile(.., q, interpolation='linear') on one ascending float32 row: rank = q*(n-1) in float32, then
    torch.lerp's two-sided formula between the neighbours."""
    n = sorted_row.shape[0]
    rank = np.float32(q) * np.float32(n - 1)
    lo = int(np.floor(rank))
    hi = int(np.ceil(rank))
    w = np.float32(rank - np.float32(lo))
    a, b = np.float32(sorted_row[lo]), np.float32(sorted_row[hi])
    if w < np.float32(0.5):
        return np.float32(a + w * np.float32(b - a))
    return np.float32(b - np.float32(b - a) * np.float32(np.float32(1.0) - w))


def lhp_depth_weights(depth: torch.Tensor, feat_hw: Tuple[int, int]):
    """The (B,P,P) propagation map of forward_depth (src/modules.py:286-319) and its per-row statistics (min, max,
    1 % quantile of the normalised distances).  Distances are the direct float32 formula sqrt(((dx^2 + dy^2) + dz^2));
    the reference's torch.cdist takes its matmul path for P > 25, whose rounding (about 1e-7 * |point|^2 on d^2, i.e. a
    non-zero self-distance) is not reproducible - the parity tests against the reference carry that tolerance."""
    pooled = adaptive_avg_pool2d(depth, tuple(feat_hw))
    b = pooled.shape[0]
    p = int(feat_hw[0]) * int(feat_hw[1])
    wmap = np.zeros((b, p, p), dtype=np.float32)
    stats = np.zeros((b, p, 3), dtype=np.float32)
    for i in range(b):
        pts = depth2points(pooled[i, 0], fov=90).reshape(3, -1).t().numpy().astype(np.float32)      # (P,3)
        dx = pts[:, None, 0] - pts[None, :, 0]
        dy = pts[:, None, 1] - pts[None, :, 1]
        dz = pts[:, None, 2] - pts[None, :, 2]
        dist = np.sqrt((dx * dx + dy * dy) + dz * dz).astype(np.float32)
        mn = dist.min(axis=1, keepdims=True)
        mx = dist.max(axis=1, keepdims=True)
        dn = ((dist - mn) / (mx - mn)).astype(np.float32)
        srt = np.sort(dn, axis=1)
        thr = np.asarray([quantile_lerp(srt[r], 0.01) for r in range(p)], dtype=np.float32)[:, None]
        wmap[i] = np.where(dn > thr, np.float32(0.0), np.float32(1.0) - dn)
        stats[i, :, 0], stats[i, :, 1], stats[i, :, 2] = mn[:, 0], mx[:, 0], thr[:, 0]
    return torch.from_numpy(wmap), torch.from_numpy(stats)


def lhp_propagate(code: torch.Tensor, depth: torch.Tensor) -> torch.Tensor:
    """code_mixed of forward_depth: out[b,:,p] = mean_q map[b,p,q] * code[b,:,q]   (src/modules.py:321-335)."""
    b, d, h, w = code.shape
    wmap, _ = lhp_depth_weights(depth, (h, w))
    flat = code.reshape(b, d, h * w)
    out = torch.einsum("bpq,bdq->bdp", wmap, flat) / float(h * w)
    return out.reshape(b, d, h, w)


def _row_quantiles(rows: np.ndarray, q: float) -> np.ndarray:
    """torch.quantile(rows, q, dim=-1, keepdim=True) on a float32 (..., n) array."""
    srt = np.sort(rows, axis=-1)
    flat = srt.reshape(-1, srt.shape[-1])
    out = np.asarray([quantile_lerp(r, q) for r in flat], dtype=np.float32)
    return out.reshape(srt.shape[:-1] + (1,))


def _heads_mean(attn: torch.Tensor) -> np.ndarray:
    """torch.mean(attn[:, :, 1:, 1:], dim=1): the heads added in order, one division (src/modules.py:239-240, 407-408)."""
    a = attn[:, :, 1:, 1:].numpy().astype(np.float32)
    acc = np.zeros_like(a[:, 0])
    for hd in range(a.shape[1]):
        acc = (acc + a[:, hd]).astype(np.float32)
    return (acc / np.float32(a.shape[1])).astype(np.float32)


def lhp_attn_weights(attn: torch.Tensor) -> torch.Tensor:
    """The (B,P,P) map of LocalHiddenPositiveProjection.forward_attn (src/modules.py:239-254): heads-mean attention between
    the patch tokens, row-wise min-max normalised, zero where above the row's 99 % quantile."""
    a = _heads_mean(attn)
    mn = a.min(axis=2, keepdims=True)
    mx = a.max(axis=2, keepdims=True)
    an = ((a - mn) / (mx - mn)).astype(np.float32)
    thr = _row_quantiles(an, 0.99)
    return torch.from_numpy(np.where(an > thr, np.float32(0.0), an))


def lhp_propagate_attn(code: torch.Tensor, attn: torch.Tensor) -> torch.Tensor:
    """code_mixed of forward_attn: out[b,:,p] = mean_q map[b,p,q] * code[b,:,q]   (src/modules.py:256-269)."""
    b, d, h, w = code.shape
    out = torch.einsum("bpq,bdq->bdp", lhp_attn_weights(attn), code.reshape(b, d, h * w)) / float(h * w)
    return out.reshape(b, d, h, w)


def lhp_index_mask(sz: int) -> np.ndarray:
    """index_mask of the LHP constructors (src/modules.py:356-383): mask[p][q] = 1 for q in the 3x3 neighbourhood of p clipped
    to the sz x sz map (the constructor's nine cases spell out exactly that)."""
    i = np.arange(sz * sz) // sz
    j = np.arange(sz * sz) % sz
    near = (np.abs(i[:, None] - i[None, :]) <= 1) & (np.abs(j[:, None] - j[None, :]) <= 1)
    return near.astype(np.float32)


def lhp_original_depth_weights(depth: torch.Tensor, sz: int) -> torch.Tensor:
    """lhp_map of OriginalLocalHiddenPositiveProjection.forward_depth (src/modules.py:441-471): 1 - normalised distance, zero
    where the distance is above the row mean, times the index mask.  Same direct distance formula as lhp_depth_weights."""
    pooled = adaptive_avg_pool2d(depth, (sz, sz))
    b, p = pooled.shape[0], sz * sz
    out = np.zeros((b, p, p), dtype=np.float32)
    mask = lhp_index_mask(sz)
    for i in range(b):
        pts = depth2points(pooled[i, 0], fov=90).reshape(3, -1).t().numpy().astype(np.float32)
        dx = pts[:, None, 0] - pts[None, :, 0]
        dy = pts[:, None, 1] - pts[None, :, 1]
        dz = pts[:, None, 2] - pts[None, :, 2]
        dist = np.sqrt((dx * dx + dy * dy) + dz * dz).astype(np.float32)
        mn = dist.min(axis=1, keepdims=True)
        mx = dist.max(axis=1, keepdims=True)
        dn = ((dist - mn) / (mx - mn)).astype(np.float32)
        mean = torch.mean(torch.from_numpy(dn), dim=1, keepdim=True).numpy()
        out[i] = np.where(dn > mean, np.float32(0.0), np.float32(1.0) - dn) * mask
    return torch.from_numpy(out)


def lhp_original_attn_weights(attn: torch.Tensor, sz: int) -> torch.Tensor:
    """attn of OriginalLocalHiddenPositiveProjection.forward_attn before the weighted sum (src/modules.py:407-418): heads-mean
    attention, (a - q10) / (q90 - q10) per row, zero below the row mean, times the index mask."""
    a = _heads_mean(attn)
    hi = _row_quantiles(a, 0.9)
    lo = _row_quantiles(a, 0.1)
    an = ((a - lo) / (hi - lo)).astype(np.float32)
    mean = torch.mean(torch.from_numpy(an), dim=2, keepdim=True).numpy()
    return torch.from_numpy(np.where(an < mean, np.float32(0.0), an) * lhp_index_mask(sz)[None])


def lhp_original_propagate(wmap: torch.Tensor, code: torch.Tensor, divide_num: torch.Tensor) -> torch.Tensor:
    """out[b,:,p] = sum_q map[b,p,q] * code[b,:,q] / divide_num[p]   (src/modules.py:420-432, 473-485); the reference's
    divide_num is all zero (re-created inside the constructor loop, :354,382 - never filled): inf / nan, as there."""
    b, d, h, w = code.shape
    s = torch.einsum("bpq,bdq->bpd", wmap, code.reshape(b, d, h * w))
    out = s / divide_num.reshape(1, h * w, 1)
    return out.permute(0, 2, 1).reshape(b, d, h, w)


def knn_table(normed_feats: torch.Tensor, k: int = 30) -> torch.Tensor:
    """Nearest-neighbour table of src/precompute_knns.py:106-112: row i = indices of the k largest entries of
    (X X^T)[i] in float32, value descending; ties by ascending index (the build's rule; torch.topk leaves it open)."""
    sims = torch.einsum("nf,mf->nm", normed_feats, normed_feats).numpy()
    order = np.lexsort((np.arange(sims.shape[1])[None, :].repeat(sims.shape[0], 0), -sims), axis=1)
    return torch.from_numpy(order[:, :k].astype(np.int64))


def topk_rows(vals: torch.Tensor, k: int) -> torch.Tensor:
    """Indices of the k largest entries per row: value descending, ties by ascending index."""
    v = vals.numpy()
    order = np.lexsort((np.arange(v.shape[1])[None, :].repeat(v.shape[0], 0), -v), axis=1)
    return torch.from_numpy(order[:, :k].astype(np.int64))


def confusion_counts(preds: torch.Tensor, target: torch.Tensor, n_classes: int, extra_clusters: int) -> torch.Tensor:
    """What one UnsupervisedMetrics.update adds to `stats` (src/utils.py:222-232): counts[pred, actual] over the elements
    with 0 <= actual < n_classes and 0 <= pred < n_classes; shape (n_classes + extra_clusters, n_classes), int64."""
    a = target.reshape(-1).numpy().astype(np.int64)
    p = preds.reshape(-1).numpy().astype(np.int64)
    ok = (a >= 0) & (a < n_classes) & (p >= 0) & (p < n_classes)
    out = np.zeros((n_classes + extra_clusters, n_classes), dtype=np.int64)
    np.add.at(out, (p[ok], a[ok]), 1)
    return torch.from_numpy(out)


# --------------------------------------------------------------------------------------
# A4  helper                                                           src/modules.py:1231-1254
# --------------------------------------------------------------------------------------
def clamp_bounds(cfg) -> Tuple[float, Optional[float]]:
    lo = 0.0 if cfg.zero_clamp else -9999.0
    hi = 0.8 if cfg.stabalize else None
    return lo, hi


def helper(cfg, f1, f2, c1, c2, shift):
    with torch.no_grad():
        fd = tensor_correlation(norm(f1), norm(f2))
        if cfg.pointwise:
            old_mean = fd.mean()
            fd = fd - fd.mean(dim=(3, 4), keepdim=True)
            fd = fd - fd.mean() + old_mean
    cd = tensor_correlation(norm(c1), norm(c2))
    lo, hi = clamp_bounds(cfg)
    loss = -cd.clamp(lo, hi) * (fd - shift)
    return loss, cd


# --------------------------------------------------------------------------------------
# A5  depth_feature_correlation                                        src/modules.py:1256-1278
# --------------------------------------------------------------------------------------
def depth_feature_correlation(cfg, c1, c2, d1, d2, shift):
    cd = tensor_correlation(norm(c1), norm(c2))
    d1 = interpolate_bilinear_ac(d1, tuple(c1.shape[2:]))
    d2 = interpolate_bilinear_ac(d2, tuple(c2.shape[2:]))
    dd = depth_correlation(norm(d1), norm(d2))
    lo, hi = clamp_bounds(cfg)
    loss = -cd.clamp(lo, hi) * (dd - shift)
    return loss, dd


# --------------------------------------------------------------------------------------
# A7  ContrastiveCorrelationLoss.forward                               src/modules.py:1280-1367
# --------------------------------------------------------------------------------------
def draw_coords(cfg, orig_feats, orig_feats_pos, depth, depth_pos, salience=None, salience_pos=None):
    """Coordinate selection (src/modules.py:1287-1321), RNG calls in the reference's order."""
    b = orig_feats.shape[0]
    s = cfg.feature_samples
    if getattr(cfg, "use_salience", False):                                        # :1290-1297
        shape = [b, s, s, 2]
        c1n = sample_nonzero_locations(salience, shape)
        c2n = sample_nonzero_locations(salience_pos, shape)
        c1r = torch.rand(shape) * 2 - 1
        c2r = torch.rand(shape) * 2 - 1
        mask = (torch.rand(shape[:-1]) > .1).unsqueeze(-1).to(torch.float32)
        return c1n * mask + c1r * (1 - mask), c2n * mask + c2r * (1 - mask)
    if cfg.depth_sampling == "simple":                                             # :1299-1302
        c1 = simple_depth_informed_sampling(tuple(orig_feats.shape[-2:]), depth, s) * 2 - 1
        c2 = simple_depth_informed_sampling(tuple(orig_feats_pos.shape[-2:]), depth_pos, s) * 2 - 1
        return c1, c2
    if cfg.depth_sampling in ("fps", "fps_depth_feat"):
        hw = tuple(orig_feats.shape[-2:])
        c1 = farthest_point_sampling_depth(hw, depth, s) * 2 - 1
        c2 = farthest_point_sampling_depth(tuple(orig_feats_pos.shape[-2:]), depth_pos, s) * 2 - 1
        return c1, c2
    c1 = torch.rand(b, s, s, 2) * 2 - 1
    c2 = torch.rand(b, s, s, 2) * 2 - 1
    return c1, c2


def forward(cfg, orig_feats, orig_feats_pos, orig_code, orig_code_pos, depth=None, depth_pos=None,
            coords1=None, coords2=None, perms: Optional[Sequence[torch.Tensor]] = None, salience=None, salience_pos=None):
    """Returns the reference's 6- or 8-tuple.  coords/perms may be injected (RNG parity)."""
    if coords1 is None or coords2 is None:
        coords1, coords2 = draw_coords(cfg, orig_feats, orig_feats_pos, depth, depth_pos, salience, salience_pos)
    feats = sample(orig_feats, coords1)
    code = sample(orig_code, coords1)
    feats_pos = sample(orig_feats_pos, coords2)
    code_pos = sample(orig_code_pos, coords2)

    pos_intra_loss, pos_intra_cd = helper(cfg, feats, feats, code, code, cfg.pos_intra_shift)
    pos_inter_loss, pos_inter_cd = helper(cfg, feats, feats_pos, code, code_pos, cfg.pos_inter_shift)
    if cfg.depth_feat_correlation_loss:
        depth_feat_loss, depth_feat_cd = depth_feature_correlation(cfg, code, code, depth, depth, cfg.depth_feat_shift)

    neg_losses, neg_cds = [], []
    for k in range(cfg.neg_samples):
        perm = perms[k] if perms is not None else super_perm(orig_feats.shape[0])
        feats_neg = sample(orig_feats[perm], coords2)
        code_neg = sample(orig_code[perm], coords2)
        l, c = helper(cfg, feats, feats_neg, code, code_neg, cfg.neg_inter_shift)
        neg_losses.append(l)
        neg_cds.append(c)
    neg_inter_loss = torch.cat(neg_losses, dim=0)
    neg_inter_cd = torch.cat(neg_cds, dim=0)

    if cfg.depth_feat_correlation_loss:
        return (pos_intra_loss.mean(), pos_intra_cd, pos_inter_loss.mean(), pos_inter_cd,
                neg_inter_loss, neg_inter_cd, depth_feat_loss.mean(), depth_feat_cd)
    return (pos_intra_loss.mean(), pos_intra_cd, pos_inter_loss.mean(), pos_inter_cd,
            neg_inter_loss, neg_inter_cd)


# --------------------------------------------------------------------------------------
# A13  caller arithmetic                                    src/train_segmentation.py:303-350
# --------------------------------------------------------------------------------------
def total_loss(cfg, out) -> torch.Tensor:
    pos_intra, pos_inter, neg = out[0].mean(), out[2].mean(), out[4].mean()
    if cfg.depth_feat_correlation_loss:
        balance = cfg.lhp_weight if (getattr(cfg, "lhp", False) and getattr(cfg, "lhp_weight_balance", False)) else 0.0
        return (cfg.pos_inter_weight * pos_inter + cfg.pos_intra_weight * pos_intra +
                cfg.neg_inter_weight * neg + cfg.depth_feat_weight * out[6].mean()) * (cfg.correspondence_weight - balance)
    return (cfg.pos_inter_weight * pos_inter + cfg.pos_intra_weight * pos_intra +
            cfg.neg_inter_weight * neg) * cfg.correspondence_weight


# --------------------------------------------------------------------------------------
# A12  decay schedules                 src/depth_decay_modules.py:4-65, train_segmentation.py:356-375
# --------------------------------------------------------------------------------------
def decay_value(kind: str, init, rate: float, update_every: int, min_value, step: int):
    k = step // update_every
    if k == 0:
        return init
    v = max(init * rate ** k, min_value) if kind == "exp" else max(init - k * rate, min_value)
    return type(init)(v) if type(v) != type(init) else v


def legacy_decay_step(cfg, loss_cfg, global_step: int) -> None:
    """Mutates cfg / loss_cfg exactly like src/train_segmentation.py:356-375 (incl. quirk Q9)."""
    if cfg.depth_loss_decay and global_step % cfg.decay_every_steps == 0 and global_step > 0:
        cfg.depth_feat_weight = cfg.depth_feat_weight * cfg.depth_loss_decay_factor
        if not cfg.fix_depth_feat_shift:
            cfg.depth_feat_shift = cfg.depth_feat_shift * cfg.depth_loss_decay_factor
    if cfg.fps_until_step > 0 and global_step >= cfg.fps_until_step:
        loss_cfg.depth_sampling = "none"
        loss_cfg.feature_samples = cfg.post_fps_samples
    if cfg.fps_sample_decay and global_step % cfg.fps_sample_decay_every_steps == 0:
        loss_cfg.feature_samples = int(loss_cfg.feature_samples * cfg.fps_sample_decay_factor)
        if loss_cfg.feature_samples < cfg.fps_min_samples:
            loss_cfg.feature_samples = cfg.fps_min_samples


# --------------------------------------------------------------------------------------
# helpers shared by tests / bench (not reference functions)
# --------------------------------------------------------------------------------------
def default_cfg(**over) -> SimpleNamespace:
    """Hot-path keys with the defaults of src/configs/local_config.yml."""
    cfg = SimpleNamespace(
        feature_samples=11, use_salience=False, depth_sampling="none", fps_gpu=False,
        pointwise=True, zero_clamp=True, stabalize=False,
        pos_intra_shift=0.08, pos_inter_shift=0.02, neg_inter_shift=0.66, neg_samples=5,
        depth_feat_correlation_loss=True, depth_feat_shift=0.03,
        pos_intra_weight=0.67, pos_inter_weight=0.25, neg_inter_weight=0.63, depth_feat_weight=0.19,
        correspondence_weight=1.0, lhp=False, lhp_weight=0.2, lhp_weight_balance=False,
        depth_loss_decay=False, depth_loss_decay_factor=1.0, decay_every_steps=300,
        fix_depth_feat_shift=False, fps_until_step=0, post_fps_samples=11, fps_sample_decay=False,
        fps_sample_decay_every_steps=300, fps_sample_decay_factor=0.9, fps_min_samples=0)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def identity_coords(b: int, s: int) -> torch.Tensor:
    """coords (B,S,S,2) such that `sample` reads pixel (y=j', x=i') ... the dense S==h grid.

    coords[b, u, v, 0] = lin[v], coords[b, u, v, 1] = lin[u]: with the permute inside `sample`
    output (i, j) reads x = coords[b, j, i, 0] = lin[i], y = coords[b, j, i, 1] = lin[j]
    -> out[b,:,i,j] = t[b,:,j,i] (an exact spatial transpose, quirk Q3).
    """
    lin = torch.linspace(-1.0, 1.0, s)
    c = torch.empty(b, s, s, 2)
    c[..., 0] = lin.view(1, 1, s)
    c[..., 1] = lin.view(1, s, 1)
    return c
