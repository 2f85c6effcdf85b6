#!/usr/bin/env python3
"""developer aid: randomized shape / flag sweep of the HIP loss against the CPU oracle (forward scalars and code gradients,
tolerances of tests/test_gpu_sweep.py).  Every case is reproducible from its seed; failures are printed and counted.
   python scripts/fuzz_parity.py [n_cases] [first_seed] [edge]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from depthg_amd import ContrastiveCorrelationLoss  # noqa: E402
from oracle import depthg_oracle as O  # noqa: E402

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
EDGE = len(sys.argv) > 3 and sys.argv[3] == "edge"       # sizes at the kernels' blocking boundaries instead of uniform draws


def pick(g, lo, hi):
    return int(torch.randint(lo, hi + 1, (), generator=g))


def one(seed):
    g = torch.Generator().manual_seed(seed)
    dense = pick(g, 0, 2) == 0
    if EDGE:
        B = [1, 2, 7, 8, 9, 16, 33, 64, 65][pick(g, 0, 8)]
        C = [31, 33, 127, 128, 129, 383, 384, 385, 767, 768, 769, 1024, 1600][pick(g, 0, 12)]     # (> 768: channel chunks on the dense grid, the small-grid kernel on <= 160 positions, refused elsewhere)
        D = [1, 7, 8, 9, 31, 32, 33, 63, 64, 65, 79, 80, 81, 95, 96, 97, 127, 128][pick(g, 0, 17)]
        N = [1, 2, 5, 7, 8][pick(g, 0, 4)]
        if dense:
            h = w = [5, 8, 11, 12, 13, 16, 17, 23, 32, 33, 40][pick(g, 0, 10)]
            if h * h * B > 40000:
                B = max(1, 40000 // (h * h))
            S = h
        else:
            h, w = pick(g, 4, 40), pick(g, 4, 40)
            S = [2, 3, 5, 8, 11, 12, 16][pick(g, 0, 6)]
            S = min(S, h, w)
            if S * S * B > 6000:
                B = max(1, 6000 // (S * S))
    else:
        B = pick(g, 1, 9)
        C = [32, 64, 100, 128, 200, 384, 384, 768][pick(g, 0, 7)]
        D = pick(g, 4, 128) if pick(g, 0, 3) else [70, 90, 96, 128][pick(g, 0, 3)]
        N = pick(g, 1, 6)                   # (the reference's torch.cat over the negatives needs at least one)
        if dense:
            h = w = pick(g, 6, 30)
            S = h
        else:
            h, w = pick(g, 5, 30), pick(g, 5, 30)
            S = pick(g, 2, min(h, w, 14))
    flags = dict(pointwise=bool(pick(g, 0, 3)), zero_clamp=bool(pick(g, 0, 3)), stabalize=pick(g, 0, 4) == 0,
                 depth_feat_correlation_loss=bool(pick(g, 0, 3)))
    nograd = pick(g, 0, 3) == 0             # forward-only call (no gradient kernels, other fused kernel and reduction)
    full = not dense and pick(g, 0, 2) == 0  # reference-shaped un-reduced outputs (dg_corr_materialize)
    line = not dense and pick(g, 0, 4) == 0  # S x 1 grids of depth_sampling='simple'
    desc = f"seed {seed}: dense={dense} B={B} C={C} D={D} {h}x{w} S={S} N={N} nograd={nograd} full={full} line={line} {flags}"
    f, fp = torch.randn(B, C, h, w, generator=g), torch.randn(B, C, h, w, generator=g)
    c, cp = torch.randn(B, D, h, w, generator=g), torch.randn(B, D, h, w, generator=g)
    d = torch.randint(0, 256, (B, 1, 3 * h + pick(g, 0, 5), 3 * w + pick(g, 0, 5)), generator=g).float()
    d[:, :, :pick(g, 0, 6), :pick(g, 0, 6)] = 0.0
    if pick(g, 0, 2) == 0 or B == 1:
        perms = [torch.randint(0, B, (B,), generator=g) for _ in range(N)]
    else:
        perms = [O.super_perm(B, g) for _ in range(N)]
    cfg = O.default_cfg(feature_samples=S, neg_samples=N, dim=D, dg_outputs="full" if full else "reduced", dg_dense_grid=dense,
                        dg_small_identity_blobs=bool(os.environ.get("DG_FUZZ_BLOBS")), **flags)   # DG_FUZZ_BLOBS=1: dense grids of <= 160 positions on the blob kernels (the route until round 6)
    if dense:
        c1 = c2 = O.identity_coords(B, h)
        kw = dict(shared_coords=True, identity_grid=True)
    else:
        shared = pick(g, 0, 2) == 0 and not full   # DG_SHARED_COORDS: ONE grid for every image and both coordinate sets
        S2 = 1 if line else S
        c1 = (torch.rand(1, S, S2, 2, generator=g).expand(B, S, S2, 2).contiguous() if shared else torch.rand(B, S, S2, 2, generator=g)) * 2.2 - 1.1
        c2 = c1 if shared else torch.rand(B, S, S2, 2, generator=g) * 2.2 - 1.1
        kw = dict(shared_coords=shared)
    cr, cpr = c.clone().requires_grad_(not nograd), cp.clone().requires_grad_(not nograd)
    ref = O.forward(cfg, f, fp, cr, cpr, d, d, coords1=c1, coords2=c2, perms=perms)
    cg, cpg = c.to(dev).requires_grad_(not nograd), cp.to(dev).requires_grad_(not nograd)
    try:
        out = ContrastiveCorrelationLoss(cfg).forward_with(f.to(dev), fp.to(dev), cg, cpg, d.to(dev), c1.to(dev), c2.to(dev),
                                                           [p.to(dev) for p in perms], **kw)
    except RuntimeError as e:
        if "not supported" in str(e) or "unsupported" in str(e).lower() or "are supported on" in str(e):     # (the stated limits: a refusal, not a result)
            return "skip", desc + f"  [{str(e)[-80:]}]"
        raise
    bad = []
    for i in range(len(ref)):
        a, b = float(out[i].detach().mean()), float(ref[i].detach().mean())
        if not abs(a - b) <= 3e-3 * abs(b) + 3e-5:
            bad.append(f"tuple[{i}] {a} vs {b}")
        if full and tuple(out[i].shape) != tuple(ref[i].shape):
            bad.append(f"tuple[{i}] shape {tuple(out[i].shape)} vs {tuple(ref[i].shape)}")
        elif full and out[i].dim() > 0:
            err = float((out[i].detach().cpu() - ref[i].detach()).abs().max())
            if not err <= 4e-3:              # cd / loss elements: fp16 code, bf16 feats on the MFMA
                bad.append(f"tuple[{i}] max element error {err:.3g}")
    if nograd:
        return ("FAIL " + "; ".join(bad), desc) if bad else ("ok", desc)
    O.total_loss(cfg, ref).backward()
    O.total_loss(cfg, out).backward()
    torch.cuda.synchronize()
    for got, want, name in ((cg.grad, cr.grad, "code"), (cpg.grad, cpr.grad, "code_pos")):
        if want is None or float(want.norm()) == 0.0:
            if got is not None and float(got.abs().max()) > 1e-12:
                bad.append(f"{name}: oracle gradient is zero, got {float(got.abs().max())}")
            continue
        if not torch.isfinite(got).all():
            bad.append(f"{name}: non-finite gradient")
            continue
        rel = float((got.cpu() - want).norm() / want.norm())
        # (D == 1: the normalised code is +-1 and its gradient vanishes identically - both sides hold rounding noise only)
        tiny = max(float(got.abs().max()), float(want.abs().max())) < (1e-6 if D == 1 else 1e-7)      # (noise at D = 1: the oracle's reached 1.4e-7, the fused small-grid kernel's 1.6e-7 - seed 9053 of the edge sweep)
        if rel > 4e-2 and not tiny:
            # what a failure is made of: the elements of the oracle's cd within 1e-3 of a clamp threshold (the fp16 cd of the
            # fused kernels may clamp those the other way: on a grid of a few positions ONE such element is percents of the
            # gradient), and - dense grid - the same call through the sampled-rows path, whose masks are exact
            thr = ([0.0] if flags["zero_clamp"] else []) + ([0.8] if flags["stabalize"] else [])
            cds = [ref[i].detach() for i in (1, 3, 5)]
            near = sum(int(((cd - t).abs() < 1e-3).sum()) for cd in cds for t in thr)
            note = f"{near} of {sum(cd.numel() for cd in cds)} cd elements within 1e-3 of a clamp threshold"
            if dense and name == "code":
                c2g = c.to(dev).requires_grad_(True)
                o2 = ContrastiveCorrelationLoss(cfg).forward_with(f.to(dev), fp.to(dev), c2g, cp.to(dev).requires_grad_(True), d.to(dev),
                                                                  c1.to(dev), c2.to(dev), [p.to(dev) for p in perms], shared_coords=True)
                O.total_loss(cfg, o2).backward()
                note += f"; through the sampled-rows path: rel L2 {float((c2g.grad.cpu() - want).norm() / want.norm()):.3g}"
            bad.append(f"{name}: rel L2 {rel:.3g} (max |want| {float(want.abs().max()):.3g}, max |got| {float(got.abs().max()):.3g}; {note})")
    return ("FAIL " + "; ".join(bad), desc) if bad else ("ok", desc)


t0 = time.time()
counts = {}
for s in range(seed0, seed0 + n_cases):
    if os.environ.get("FUZZ_TRACE"):      # (a GPU fault kills the process: name the case first)
        print("seed", s, flush=True)
    try:
        status, desc = one(s)
    except Exception as e:  # noqa: BLE001
        status, desc = f"ERROR {type(e).__name__}: {str(e)[:300]}", f"seed {s}"
    counts[status.split()[0]] = counts.get(status.split()[0], 0) + 1
    if status != "ok":
        print(status, "|", desc, flush=True)
print(f"{n_cases} cases in {time.time() - t0:.0f} s:", counts)
