"""Golden vectors for the salience / 'simple' samplers and the loss on their coordinates,
captured by IMPORTING the reference on CPU (build container only; see make_fixtures.py for the import recipe).

    python tests/golden/make_sampler_fixtures.py     # writes samplers.npz, forward_salience.npz, forward_simple.npz

Sampler outputs depend on the global torch CPU RNG: each record stores the seed set right before the call, and the
oracle - which issues the same RNG calls in the same order - must reproduce the coordinates exactly.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_fixtures as mf  # noqa: E402

OUT = mf.OUT


def run_forward_capture(M, cfg, f, fp, c, cp, d, dp, sal, salp, rng_seed):
    """Like make_fixtures.run_reference_forward, but the coords are captured at M.sample (works for every sampler)."""
    seen, perms = [], []
    orig_sample, orig_sp = M.sample, M.super_perm

    def sample_wrap(t, coords):
        seen.append(coords.clone())
        return orig_sample(t, coords)

    def sp_wrap(size, device):
        p = orig_sp(size, device)
        perms.append(p.clone())
        return p

    c = c.clone().requires_grad_(True)
    cp = cp.clone().requires_grad_(True)
    torch.manual_seed(rng_seed)
    M.sample, M.super_perm = sample_wrap, sp_wrap
    try:
        out = M.ContrastiveCorrelationLoss(cfg)(f, fp, sal, salp, c, cp, d, dp)
    finally:
        M.sample, M.super_perm = orig_sample, orig_sp
    coords1, coords2 = seen[0], seen[2]          # sample(feats, coords1), sample(code, coords1), sample(feats_pos, coords2), ...
    intra, inter, neg = out[0].mean(), out[2].mean(), out[4].mean()
    total = (cfg.pos_inter_weight * inter + cfg.pos_intra_weight * intra + cfg.neg_inter_weight * neg +
             cfg.depth_feat_weight * out[6].mean()) * cfg.correspondence_weight
    total.backward()
    res = {k: np.asarray(getattr(cfg, k)) for k in mf.CFG_KEYS}
    res.update(use_salience=np.asarray(bool(cfg.use_salience)), rng_seed=np.asarray(rng_seed),
               feats=f.numpy(), feats_pos=fp.numpy(), code=c.detach().numpy(), code_pos=cp.detach().numpy(),
               depth=d.numpy(), depth_pos=dp.numpy(), coords1=coords1.numpy(), coords2=coords2.numpy(),
               perms=torch.stack(perms).numpy(),
               pos_intra_loss=out[0].detach().numpy(), pos_inter_loss=out[2].detach().numpy(),
               neg_inter_loss_mean=out[4].mean().detach().numpy(),
               pos_intra_cd_mean=out[1].mean().detach().numpy(), pos_inter_cd_mean=out[3].mean().detach().numpy(),
               neg_inter_cd_mean=out[5].mean().detach().numpy(),
               depth_feat_loss=out[6].detach().numpy(), depth_feat_cd_mean=out[7].mean().detach().numpy(),
               total=total.detach().numpy(), grad_code=c.grad.numpy(), grad_code_pos=cp.grad.numpy(),
               store_full=np.asarray(True), sub=np.asarray(1))
    for name, t in zip(["pos_intra_cd", "pos_inter_cd", "neg_inter_loss", "neg_inter_cd", "depth_feat_cd"],
                       [out[1], out[3], out[4], out[5], out[7]]):
        res[name] = t.detach().numpy()
    if sal is not None:
        res.update(salience=sal.numpy(), salience_pos=salp.numpy())
    return res


def main():
    M, _ = mf.import_reference()
    torch.set_num_threads(4)
    g = torch.Generator().manual_seed(77)
    fx = {}

    # ---- sample_nonzero_locations: non-square map (the reference divides both coordinates by the HEIGHT), one image
    #      without non-zeros (randint fallback), one with a single non-zero
    sal = (torch.rand(4, 24, 20, generator=g) > 0.7).float() * torch.rand(4, 24, 20, generator=g)
    sal[1] = 0.0
    sal[2] = 0.0
    sal[2, 17, 3] = 2.5
    fx["sal_map"] = sal.numpy()
    for seed, S in ((11, 5), (12, 3)):
        torch.manual_seed(seed)
        fx[f"sal_coords_seed{seed}_S{S}"] = M.sample_nonzero_locations(sal, [4, S, S, 2]).numpy()

    # ---- simple_depth_informed_sampling: few distinct values (long runs), 8-bit depths (short runs), float depths with
    #      non-divisible pooling windows, negative values / signed zero
    feat = torch.zeros(3, 4, 14, 14)
    d_runs = torch.randint(0, 4, (3, 1, 56, 56), generator=g).float() * 0.26
    d_8bit = torch.randint(0, 256, (3, 1, 112, 112), generator=g).float()
    d_float = torch.rand(3, 1, 50, 45, generator=g) * 3 - 1.5
    d_float[0, 0, :9, :9] = -0.04                      # rounds to -0.0
    for name, dm, n, seed, hw in (("runs", d_runs, 9, 21, (14, 14)), ("8bit", d_8bit, 11, 22, (14, 14)),
                                  ("float", d_float, 6, 23, (14, 14)), ("rect", d_float, 5, 24, (7, 10))):
        torch.manual_seed(seed)
        t = torch.zeros(3, 4, hw[0], hw[1])
        fx[f"simple_{name}_depth"] = dm.numpy()
        fx[f"simple_{name}_hw"] = np.asarray(hw)
        fx[f"simple_{name}_n"] = np.asarray(n)
        fx[f"simple_{name}_seed"] = np.asarray(seed)
        fx[f"simple_{name}_coords"] = M.simple_depth_informed_sampling(t, dm, n).numpy()
        fx[f"simple_{name}_pool"] = torch.nn.functional.adaptive_max_pool2d(dm, hw).numpy()
    np.savez_compressed(os.path.join(OUT, "samplers.npz"), **fx)

    # ---- whole forward on these samplers' coordinates
    # (sizes like config 1: on a much smaller problem a single cd ~ 0 element whose clamp mask flips under the fp16
    #  operands of the HIP path moves the inter-pair gradient by several per cent)
    f, fp, c, cp, d, dp = mf.gen_inputs(4321, 3, 64, 70, 14, 14, 112, zero_frac=0.02)
    cfg = mf.make_cfg(use_salience=True, feature_samples=10, neg_samples=3)
    salm = (torch.rand(3, 112, 112, generator=g) > 0.8).float()
    salm[2] = 0.0
    salp = (torch.rand(3, 112, 112, generator=g) > 0.5).float()
    res = run_forward_capture(M, cfg, f, fp, c, cp, d, dp, salm, salp, rng_seed=31)
    np.savez_compressed(os.path.join(OUT, "forward_salience.npz"), **res)
    print("salience total", float(res["total"]), res["coords1"].shape)

    d2 = (d / 16).round()          # few distinct depth values: the two-stage draw has real runs
    dp2 = (dp / 16).round()
    cfg = mf.make_cfg(depth_sampling="simple", feature_samples=9, neg_samples=3)
    res = run_forward_capture(M, cfg, f, fp, c, cp, d2, dp2, None, None, rng_seed=32)
    np.savez_compressed(os.path.join(OUT, "forward_simple.npz"), **res)
    print("simple total", float(res["total"]), res["coords1"].shape, res["pos_intra_cd"].shape)


if __name__ == "__main__":
    main()
