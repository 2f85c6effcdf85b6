#!/usr/bin/env python3
"""developer aid: randomized sweep of the samplers and the LHP propagation against the CPU oracle (bit-exact where the tests
demand it).  python scripts/fuzz_samplers.py [n_cases] [first_seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from depthg_amd import ops  # noqa: E402
from oracle import depthg_oracle as O  # noqa: E402

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def pick(g, lo, hi):
    return int(torch.randint(lo, hi + 1, (), generator=g))


def depth_map(g, B, H, W):
    kind = pick(g, 0, 3)
    if kind == 0:
        d = torch.rand(B, 1, H, W, generator=g) * 9 + 0.5
    elif kind == 1:
        d = torch.randint(0, 256, (B, 1, H, W), generator=g).float()
    elif kind == 2:
        d = torch.round(torch.rand(B, 1, H, W, generator=g) * 3)               # few values, zeros included
    else:
        y, x = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
        d = (40 + 100 * x + 60 * y).round().expand(B, 1, H, W).clone()
    if pick(g, 0, 1):
        d[:, :, : pick(g, 0, H // 2), : pick(g, 0, W // 2)] = float(pick(g, 0, 4))
    return d


def fps_case(seed):
    g = torch.Generator().manual_seed(seed)
    h, w = pick(g, 3, 40), pick(g, 3, 40)
    if pick(g, 0, 1):
        w = h
    H, W = h * pick(g, 1, 9) + pick(g, 0, h - 1), w * pick(g, 1, 9) + pick(g, 0, w - 1)
    S = pick(g, 1, min(16, int((h * w) ** 0.5)))
    B = pick(g, 1, 4)
    d = depth_map(g, B, H, W)
    desc = f"fps seed {seed}: B={B} depth {H}x{W} -> {h}x{w}, S={S}"
    want_c, want_i = O.farthest_point_sampling_depth((h, w), d, S, return_inds=True)
    got_c, got_i = ops.fps_coords(d.to(dev), (h, w), S, return_inds=True)
    if not np.array_equal(got_i.cpu().numpy(), np.asarray(want_i)):
        k = int(np.argmax((got_i.cpu().numpy() != np.asarray(want_i)).reshape(B, -1).any(0)))
        return f"FAIL selection order differs from sample {k}", desc
    if not np.array_equal(got_c.cpu().numpy(), (want_c * 2 - 1).numpy()):
        return "FAIL coords differ", desc
    return "ok", desc


def lhp_case(seed):
    g = torch.Generator().manual_seed(seed)
    sz = pick(g, 3, 30)
    B, D = pick(g, 1, 3), pick(g, 1, 128)
    H = sz * pick(g, 1, 8) + pick(g, 0, sz - 1)
    code = torch.randn(B, D, sz, sz, generator=g)
    d = depth_map(g, B, H, H)
    d = d + torch.rand(d.shape, generator=g) * 0.37 + 0.1       # distinct positive depths (equal points give 0/0 rows in the reference too)
    desc = f"lhp seed {seed}: B={B} D={D} {sz}x{sz} depth {H}"
    out, points, stats = ops.lhp_forward(code.to(dev), d.to(dev))
    wmap, ostats = O.lhp_depth_weights(d, (sz, sz))
    bad = []
    if not torch.equal(stats.cpu(), ostats):
        bad.append(f"row statistics differ in {int((stats.cpu() != ostats).any(-1).sum())} rows")
    want = O.lhp_propagate(code, d)
    rel = float((out.cpu() - want).norm() / want.norm())
    if not rel < 2e-6:
        bad.append(f"propagated code rel {rel:.3g}")
    return ("FAIL " + "; ".join(bad), desc) if bad else ("ok", desc)


def lhp_maps_case(seed):
    """dg_lhp_map_forward / backward, all three modes (attention strategy, the Original class) against the oracle."""
    from depthg_amd.lhp import neighbour_counts
    g = torch.Generator().manual_seed(seed)
    sz = pick(g, 3, 24)
    B, D, heads = pick(g, 1, 2), pick(g, 1, 128), pick(g, 1, 12)
    P = sz * sz
    code = torch.randn(B, D, sz, sz, generator=g)
    attn = torch.softmax((0.5 + 2 * torch.rand((), generator=g)) * torch.randn(B, heads, P + 1, P + 1, generator=g), dim=-1)
    H = sz * pick(g, 1, 6) + pick(g, 0, sz - 1)
    d = depth_map(g, B, H, H) + torch.rand(B, 1, H, H, generator=g) * 0.37 + 0.1
    up = torch.randn(B, D, sz, sz, generator=g)
    counts = neighbour_counts(sz)
    desc = f"lhp maps seed {seed}: B={B} D={D} heads={heads} {sz}x{sz} depth {H}"
    bad = []
    out, wmap = ops.lhp_map_forward(ops.LHP_ATTN, code.to(dev), attn=attn.to(dev))
    omap = O.lhp_attn_weights(attn)
    if not torch.equal(wmap.cpu(), omap):
        bad.append(f"attention map differs in {int((wmap.cpu() != omap).sum())} entries")
    want = O.lhp_propagate_attn(code, attn)
    rel = float((out.cpu() - want).norm() / want.norm())
    if not rel < 3e-6:
        bad.append(f"attn forward rel {rel:.3g}")
    gb = ops.lhp_map_backward(ops.LHP_ATTN, up.to(dev), wmap).cpu()
    gw = (torch.einsum("bpq,bdp->bdq", omap, up.reshape(B, D, P)) / float(P)).reshape(B, D, sz, sz)
    rel = float((gb - gw).norm() / gw.norm())
    if not rel < 3e-6:
        bad.append(f"attn backward rel {rel:.3g}")
    for mode, kw, om in ((ops.LHP_ORIG_ATTN, dict(attn=attn.to(dev)), O.lhp_original_attn_weights(attn, sz)),
                         (ops.LHP_ORIG_DEPTH, dict(depth=d.to(dev)), O.lhp_original_depth_weights(d, sz))):
        out, w9 = ops.lhp_map_forward(mode, code.to(dev), divide=counts.float().to(dev), **kw)
        want = O.lhp_original_propagate(om, code, counts)
        rows = int(((out.cpu() - want).abs() > 2e-5 * (1 + want.abs())).reshape(B, D, P).any(1).sum())
        if rows > 2:                     # (a weight at the row-mean threshold may fall the other way)
            bad.append(f"mode {mode}: {rows} output positions differ")
        gb = ops.lhp_map_backward(mode, up.to(dev), w9, counts.float().to(dev)).cpu()
        gw = torch.einsum("bpq,bdp->bdq", om, up.reshape(B, D, P) / counts.reshape(1, 1, P)).reshape(B, D, sz, sz)
        rows = int(((gb - gw).abs() > 2e-5 * (1 + gw.abs())).reshape(B, D, P).any(1).sum())
        if rows > 18:
            bad.append(f"mode {mode}: {rows} gradient positions differ")
    return ("FAIL " + "; ".join(bad), desc) if bad else ("ok", desc)


def edge_uniforms(u, g):
    """uniforms with the ends of [0, 1) forced in: rank 0 and the last rank get drawn"""
    u = u.clone()
    flat = u.view(-1)
    if flat.numel() >= 2:
        flat[pick(g, 0, flat.numel() - 1)] = 0.0
        flat[pick(g, 0, flat.numel() - 1)] = float(np.nextafter(np.float32(1.0), np.float32(0.0)))
    return u


def salience_case(seed):
    g = torch.Generator().manual_seed(seed)
    B, H, W, S = pick(g, 1, 4), pick(g, 3, 260), pick(g, 3, 330), pick(g, 1, 13)
    dens = [0.0, 0.003, 0.05, 0.5, 1.0][pick(g, 0, 4)]
    sal = (torch.rand(B, H, W, generator=g) < dens).float() * float(pick(g, 1, 3))
    if pick(g, 0, 2) == 0:
        sal[pick(g, 0, B - 1)] = 0.0
    u = edge_uniforms(torch.rand(B, S * S, generator=g), g)
    ufb = edge_uniforms(torch.rand(B, S * S, 2, generator=g), g)
    desc = f"salience seed {seed}: B={B} {H}x{W} S={S} density {dens}"
    got = ops.salience_coords(sal.to(dev), S, u.to(dev), ufb.to(dev)).cpu()
    want = O.sample_nonzero_locations_from_uniform(sal, [B, S, S, 2], u.numpy(), ufb.numpy())
    return ("ok", desc) if got.shape == want.shape and torch.equal(got, want) else ("FAIL coords differ", desc)


def simple_case(seed):
    g = torch.Generator().manual_seed(seed)
    h, w = pick(g, 2, 64), pick(g, 2, 64)
    H, W = h * pick(g, 1, 6) + pick(g, 0, h - 1), w * pick(g, 1, 6) + pick(g, 0, w - 1)
    B, n = pick(g, 1, 3), pick(g, 1, 40)
    d = depth_map(g, B, H, W)
    uv = edge_uniforms(torch.rand(B, n, generator=g), g)
    up = edge_uniforms(torch.rand(B, n, generator=g), g)
    desc = f"simple seed {seed}: B={B} depth {H}x{W} -> {h}x{w}, n={n}"
    got = ops.simple_depth_coords(d.to(dev), (h, w), n, uv.to(dev), up.to(dev)).cpu()
    want = O.simple_depth_informed_sampling_from_uniform((h, w), d, n, uv.numpy(), up.numpy()) * 2 - 1
    return ("ok", desc) if got.shape == (B, n, 1, 2) and torch.equal(got, want) else ("FAIL coords differ", desc)


def topk_case(seed):
    g = torch.Generator().manual_seed(seed)
    rows, cols = pick(g, 1, 40), [pick(g, 1, 70), pick(g, 65, 3000), pick(g, 3000, 60000)][pick(g, 0, 2)]
    k = pick(g, 1, min(64, cols))
    kind = pick(g, 0, 2)
    m = torch.randn(rows, cols, generator=g) if kind == 0 else \
        (torch.randint(0, pick(g, 1, 5) + 1, (rows, cols), generator=g).float() - 1.5 if kind == 1 else torch.zeros(rows, cols))
    desc = f"topk seed {seed}: {rows}x{cols} k={k} kind={kind}"
    idx, val = ops.topk_rows(m.to(dev), k, return_values=True)
    want = O.topk_rows(m, k)
    ok = torch.equal(idx.cpu(), want) and torch.equal(val.cpu(), torch.gather(m, 1, want))
    return ("ok", desc) if ok else ("FAIL indices / values differ", desc)


def confusion_case(seed):
    from depthg_amd.metrics import UnsupervisedMetrics
    g = torch.Generator().manual_seed(seed)
    n, e = pick(g, 2, 200), pick(g, 0, 12)
    m = UnsupervisedMetrics("t/", n, e, True)
    want = np.zeros((n + e, n), dtype=np.int64)
    desc = f"confusion seed {seed}: n={n} extra={e}"
    for _ in range(pick(g, 1, 3)):
        shape = (pick(g, 1, 6), pick(g, 1, 200), pick(g, 1, 200))
        target = torch.randint(-1, n + 2, shape, generator=g)
        target[target == n + 1] = 255
        preds = torch.randint(-1, n + e + 2, shape, generator=g)
        m.update(preds.to(dev), target.to(dev))
        t, p_ = target.reshape(-1).numpy(), preds.reshape(-1).numpy()
        ok = (t >= 0) & (t < n) & (p_ >= 0) & (p_ < n)       # src/utils.py:222-232: both are masked with n_classes
        np.add.at(want, (p_[ok], t[ok]), 1)
    return ("ok", desc) if np.array_equal(m.stats.cpu().numpy(), want) else ("FAIL stats differ", desc)


t0 = time.time()
counts = {}
for s in range(seed0, seed0 + n_cases):
    for fn in (fps_case, lhp_case, lhp_maps_case, salience_case, simple_case, topk_case, confusion_case):
        try:
            status, desc = fn(s)
        except Exception as e:  # noqa: BLE001
            status, desc = f"ERROR {type(e).__name__}: {str(e)[:300]}", f"{fn.__name__} seed {s}"
        key = fn.__name__ + " " + status.split()[0]
        counts[key] = counts.get(key, 0) + 1
        if status != "ok":
            print(status, "|", desc, flush=True)
print(f"{n_cases} cases in {time.time() - t0:.0f} s:", counts)
