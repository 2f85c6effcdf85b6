"""Generate golden vectors by IMPORTING the reference on CPU (build container only).

    python tests/golden/make_fixtures.py            # rewrites tests/golden/*.npz

The reference lives read-only at /root/reference and cannot travel to the GPU box, so its
outputs on seeded inputs are captured here as data (inputs + expected outputs); no reference
source is copied.  Import recipe: SURVEY.md section 8(c) (four stub modules).
RNG-dependent pieces (torch.rand coords, super_perm) are captured by wrapping them while the
reference runs, and stored as explicit `coords1/coords2/perms` in every forward fixture.
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    sys.dont_write_bytecode = True
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    stub("wget")
    class Metric:  # torchmetrics.Metric stand-in (src/utils.py:16,202): states are plain attributes
        def __init__(self, *a, **k):
            pass

        def add_state(self, name, default, dist_reduce_fx=None):
            setattr(self, name, default.clone())
    stub("torchmetrics", Metric=Metric)
    tv = stub("torchvision")
    tv.models = stub("torchvision.models")
    class _Any:
        def __init__(self, *a, **k):
            pass
    tv.transforms = stub("torchvision.transforms", Normalize=_Any)
    tb = stub("torch.utils.tensorboard")
    tb.summary = stub("torch.utils.tensorboard.summary", hparams=lambda *a, **k: None)
    sys.path.insert(0, os.path.join(REF, "src"))
    import modules as M  # noqa
    import depth_decay_modules as DD  # noqa
    return M, DD


def make_cfg(**over):
    cfg = SimpleNamespace(
        feature_samples=11, use_salience=False, depth_sampling="none", fps_gpu=False,
        pointwise=True, zero_clamp=True, stabalize=False,
        pos_intra_shift=0.08, pos_inter_shift=0.02, neg_inter_shift=0.66, neg_samples=5,
        depth_feat_correlation_loss=True, depth_feat_shift=0.03,
        pos_intra_weight=0.67, pos_inter_weight=0.25, neg_inter_weight=0.63, depth_feat_weight=0.19,
        correspondence_weight=1.0)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


CFG_KEYS = ["feature_samples", "depth_sampling", "pointwise", "zero_clamp", "stabalize", "pos_intra_shift",
            "pos_inter_shift", "neg_inter_shift", "neg_samples", "depth_feat_correlation_loss",
            "depth_feat_shift", "pos_intra_weight", "pos_inter_weight", "neg_inter_weight",
            "depth_feat_weight", "correspondence_weight"]


def gen_inputs(seed, B, C, D, h, w, Himg, zero_frac=0.0, corr=0.0):
    g = torch.Generator().manual_seed(seed)
    f = torch.randn(B, C, h, w, generator=g)
    fp = torch.randn(B, C, h, w, generator=g)
    c = torch.randn(B, D, h, w, generator=g)
    cp = torch.randn(B, D, h, w, generator=g)
    d = torch.randint(0, 256, (B, 1, Himg, Himg), generator=g).float()
    dp = torch.randint(0, 256, (B, 1, Himg, Himg), generator=g).float()
    if corr > 0:  # correlated features: common component (near-cancelling neg term)
        common = torch.randn(1, C, 1, 1, generator=g)
        f = f + corr * common
        fp = fp + corr * common
    if zero_frac > 0:
        m = torch.rand(B, 1, Himg, Himg, generator=g) < zero_frac
        d = torch.where(m, torch.zeros_like(d), d)
        # also a whole zero block so that the S x S resize hits exact zeros
        d[:, :, : Himg // 4, : Himg // 3] = 0.0
    return f, fp, c, cp, d, dp


def run_reference_forward(M, cfg, f, fp, c, cp, d, dp, rng_seed=0, store_full=True, sub=37):
    """Runs M.ContrastiveCorrelationLoss and captures coords/perms; returns dict of arrays."""
    rec = {"rand": [], "perm": []}
    orig_rand, orig_sp, orig_fpsd = torch.rand, M.super_perm, M.farthest_point_sampling_depth

    def rand_wrap(*a, **k):
        r = orig_rand(*a, **k)
        rec["rand"].append(r.clone())
        return r

    def sp_wrap(size, device):
        p = orig_sp(size, device)
        rec["perm"].append(p.clone())
        return p

    fps_coords = []

    def fpsd_wrap(*a, **k):
        r = orig_fpsd(*a, **k)
        fps_coords.append(r.clone())
        return r

    c = c.clone().requires_grad_(True)
    cp = cp.clone().requires_grad_(True)
    torch.manual_seed(rng_seed)
    torch.rand = rand_wrap
    M.super_perm = sp_wrap
    M.farthest_point_sampling_depth = fpsd_wrap
    try:
        loss_fn = M.ContrastiveCorrelationLoss(cfg)
        use_depth = cfg.depth_feat_correlation_loss
        out = loss_fn(f, fp, None, None, c, cp, d if use_depth else None, dp if use_depth else None)
    finally:
        torch.rand = orig_rand
        M.super_perm = orig_sp
        M.farthest_point_sampling_depth = orig_fpsd
    if cfg.depth_sampling in ("fps", "fps_depth_feat"):
        coords1, coords2 = fps_coords[0] * 2 - 1, fps_coords[1] * 2 - 1
    else:
        coords1, coords2 = rec["rand"][0] * 2 - 1, rec["rand"][1] * 2 - 1
    perms = torch.stack(rec["perm"]) if rec["perm"] else torch.zeros(0, f.shape[0], dtype=torch.long)

    # caller arithmetic, src/train_segmentation.py:303-350 (restated; file not importable here)
    intra, inter, neg = out[0].mean(), out[2].mean(), out[4].mean()
    total = cfg.pos_inter_weight * inter + cfg.pos_intra_weight * intra + cfg.neg_inter_weight * neg
    if cfg.depth_feat_correlation_loss:
        total = total + cfg.depth_feat_weight * out[6].mean()
    total = total * cfg.correspondence_weight
    total.backward()

    res = {k: np.asarray(getattr(cfg, k)) for k in CFG_KEYS}
    res.update(feats=f.numpy(), feats_pos=fp.numpy(), code=c.detach().numpy(), code_pos=cp.detach().numpy(),
               depth=d.numpy(), depth_pos=dp.numpy(), coords1=coords1.numpy(), coords2=coords2.numpy(),
               perms=perms.numpy(),
               pos_intra_loss=out[0].detach().numpy(), pos_inter_loss=out[2].detach().numpy(),
               neg_inter_loss_mean=out[4].mean().detach().numpy(),
               pos_intra_cd_mean=out[1].mean().detach().numpy(), pos_inter_cd_mean=out[3].mean().detach().numpy(),
               neg_inter_cd_mean=out[5].mean().detach().numpy(),
               total=total.detach().numpy(), grad_code=c.grad.numpy(), grad_code_pos=cp.grad.numpy(),
               store_full=np.asarray(store_full), sub=np.asarray(sub))
    names = ["pos_intra_cd", "pos_inter_cd", "neg_inter_loss", "neg_inter_cd"]
    tens = [out[1], out[3], out[4], out[5]]
    if cfg.depth_feat_correlation_loss:
        res.update(depth_feat_loss=out[6].detach().numpy(), depth_feat_cd_mean=out[7].mean().detach().numpy())
        names.append("depth_feat_cd")
        tens.append(out[7])
    for n, t in zip(names, tens):
        flat = t.detach().reshape(-1)
        res[n] = t.detach().numpy() if store_full else flat[::sub].numpy()
    return res


def main():
    M, DD = import_reference()
    torch.set_num_threads(4)

    # ------------------------------------------------------------------ per-function fixtures
    g = torch.Generator().manual_seed(7)
    fx = {}
    t = torch.randn(2, 16, 5, 6, generator=g)
    t[0, :, 1, 2] = 0.0                      # zero vector -> eps path
    t[1, :, 0, 0] *= 1e-12
    fx["norm_in"], fx["norm_out"] = t.numpy(), M.norm(t).numpy()
    a = torch.randn(2, 8, 3, 4, generator=g)
    b = torch.randn(2, 8, 5, 2, generator=g)
    fx["corr_a"], fx["corr_b"], fx["corr_out"] = a.numpy(), b.numpy(), M.tensor_correlation(a, b).numpy()
    # sample: interior, exact border, out of range coords
    src = torch.randn(2, 5, 7, 9, generator=g)
    co = torch.rand(2, 4, 4, 2, generator=g) * 2 - 1
    co[0, 0, 0] = torch.tensor([-1.0, -1.0]); co[0, 0, 1] = torch.tensor([1.0, 1.0])
    co[0, 1, 0] = torch.tensor([1.3, -1.2]);  co[0, 1, 1] = torch.tensor([0.0, 1.0])
    fx["sample_t"], fx["sample_coords"], fx["sample_out"] = src.numpy(), co.numpy(), M.sample(src, co).numpy()
    # non-square sample grid is not used by the reference (S x S); FPS-style coords:
    dmap = torch.randint(0, 256, (3, 1, 112, 112), generator=g).float()
    dmap[1, :, :40, :50] = 0.0
    featmap = torch.zeros(3, 4, 14, 14)
    for S in (6, 11):
        coords = M.farthest_point_sampling_depth(featmap, dmap, S)
        fx[f"fpsd_coords_S{S}"] = coords.numpy()
    fx["fpsd_depth"] = dmap.numpy()
    # non-divisible pooling (100 -> 14) and float depths
    dmap2 = torch.rand(2, 1, 100, 100, generator=g) * 10
    fx["fpsd2_depth"] = dmap2.numpy()
    fx["fpsd2_coords_S5"] = M.farthest_point_sampling_depth(featmap[:2], dmap2, 5).numpy()
    fx["pool2_out"] = torch.nn.functional.adaptive_avg_pool2d(dmap2, (14, 14)).numpy()
    # depth2points + fps index order
    dsmall = torch.nn.functional.adaptive_avg_pool2d(dmap, (14, 14))[0, 0]
    pts = M.depth2points(dsmall, fov=90)
    fx["d2p_depth"], fx["d2p_out"] = dsmall.numpy(), pts.numpy()
    pc = pts.permute(1, 2, 0).reshape(-1, 3)
    _, inds = M.fps(pc, 36)
    fx["fps_points"], fx["fps_inds36"] = pc.numpy(), np.asarray(inds)
    # ties: constant depth plane (many equal distances)
    flat = torch.full((14, 14), 3.0)
    pcf = M.depth2points(flat, fov=90).permute(1, 2, 0).reshape(-1, 3)
    _, indsf = M.fps(pcf, 25)
    fx["fps_points_flat"], fx["fps_inds_flat25"] = pcf.numpy(), np.asarray(indsf)
    fx["fov_factor"] = (2.0 * torch.tan(torch.tensor([90.0]) / 2.0)).numpy()
    # interpolate (depth term)
    fx["interp_in"] = dmap.numpy()
    fx["interp_out_11"] = torch.nn.functional.interpolate(dmap, size=(11, 11), mode="bilinear", align_corners=True).numpy()
    # super_perm mapping from a given randperm
    torch.manual_seed(3)
    fx["superperm_rng_state_seed"] = np.asarray(3)
    fx["superperm_8"] = M.super_perm(8, torch.device("cpu")).numpy()
    torch.manual_seed(3)
    fx["randperm_8"] = torch.randperm(8).numpy()
    fx["superperm_1"] = M.super_perm(1, torch.device("cpu")).numpy()
    # helper / depth_feature_correlation direct
    cfg = make_cfg()
    lf = M.ContrastiveCorrelationLoss(cfg)
    f1 = torch.randn(2, 32, 4, 4, generator=g); f2 = torch.randn(2, 32, 4, 4, generator=g)
    c1 = torch.randn(2, 10, 4, 4, generator=g); c2 = torch.randn(2, 10, 4, 4, generator=g)
    for name, over in (("pw", {}), ("nopw", {"pointwise": False}), ("nozc", {"zero_clamp": False}),
                       ("stab", {"stabalize": True})):
        lf.cfg = make_cfg(**over)
        l, cdv = lf.helper(f1, f2, c1, c2, 0.3)
        fx[f"helper_{name}_loss"], fx[f"helper_{name}_cd"] = l.numpy(), cdv.numpy()
    fx["helper_f1"], fx["helper_f2"], fx["helper_c1"], fx["helper_c2"] = f1.numpy(), f2.numpy(), c1.numpy(), c2.numpy()
    lf.cfg = make_cfg()
    dd1 = torch.randint(0, 3, (2, 1, 20, 20), generator=g).float()
    l, ddv = lf.depth_feature_correlation(c1, c1, dd1, dd1, 0.03)
    fx["dfc_depth"], fx["dfc_loss"], fx["dfc_dd"] = dd1.numpy(), l.numpy(), ddv.numpy()
    np.savez_compressed(os.path.join(OUT, "functions.npz"), **fx)

    # ------------------------------------------------------------------ whole-forward fixtures
    cases = {
        # config 1 of BASELINE.json: B=2, C=64, h=w=14 (+ D=70, S=11)
        "c1_none": dict(shape=(2, 64, 70, 14, 14, 112), cfg={}, full=True),
        "c1_fps": dict(shape=(2, 64, 70, 14, 14, 112), cfg={"depth_sampling": "fps"}, full=True),
        "nopointwise": dict(shape=(2, 48, 20, 8, 8, 64), cfg={"pointwise": False, "feature_samples": 6}, full=True),
        "nozeroclamp": dict(shape=(2, 48, 20, 8, 8, 64), cfg={"zero_clamp": False, "feature_samples": 6}, full=True),
        "stabalize": dict(shape=(2, 48, 20, 8, 8, 64), cfg={"stabalize": True, "feature_samples": 6,
                                                              "neg_inter_shift": 0.1}, full=True),
        "nodepthloss": dict(shape=(2, 48, 20, 8, 8, 64), cfg={"depth_feat_correlation_loss": False,
                                                                "feature_samples": 6}, full=True),
        "zerodepth_fps": dict(shape=(3, 32, 24, 14, 14, 112), cfg={"depth_sampling": "fps", "feature_samples": 7},
                              zero_frac=0.02, full=True),
        "batch1": dict(shape=(1, 32, 24, 10, 10, 80), cfg={"feature_samples": 5}, full=True),
        "S9": dict(shape=(2, 96, 90, 14, 14, 112), cfg={"feature_samples": 9, "depth_sampling": "fps"}, full=False),
        "S12": dict(shape=(2, 96, 90, 28, 28, 224), cfg={"feature_samples": 12, "depth_sampling": "fps",
                                                           "neg_samples": 3}, full=False),
        "S14_dim100": dict(shape=(2, 128, 100, 14, 14, 112), cfg={"feature_samples": 14, "pointwise": False,
                                                                   "neg_samples": 2}, full=False),
        "corr_feats": dict(shape=(4, 64, 70, 14, 14, 112), cfg={"feature_samples": 8}, corr=1.5, full=False),
    }
    for i, (name, spec) in enumerate(cases.items()):
        B, C, D, h, w, Himg = spec["shape"]
        f, fp, c, cp, d, dp = gen_inputs(1234 + i, B, C, D, h, w, Himg, spec.get("zero_frac", 0.0), spec.get("corr", 0.0))
        cfg = make_cfg(**spec["cfg"])
        res = run_reference_forward(M, cfg, f, fp, c, cp, d, dp, rng_seed=i, store_full=spec["full"])
        np.savez_compressed(os.path.join(OUT, f"forward_{name}.npz"), **res)
        print(name, "total", float(res["total"]), "intra", float(res["pos_intra_loss"]),
              "|gc|", float(np.linalg.norm(res["grad_code"])))

    # the survey's recorded known answers re-derived: recipe seed 1234
    g = torch.Generator().manual_seed(1234)
    f = torch.randn(2, 64, 14, 14, generator=g); fp = torch.randn(2, 64, 14, 14, generator=g)
    c = torch.randn(2, 70, 14, 14, generator=g); cp = torch.randn(2, 70, 14, 14, generator=g)
    d = torch.randint(0, 256, (2, 1, 112, 112), generator=g).float()
    dp = torch.randint(0, 256, (2, 1, 112, 112), generator=g).float()
    for mode in ("none", "fps"):
        res = run_reference_forward(M, make_cfg(depth_sampling=mode), f, fp, c, cp, d, dp, rng_seed=0, store_full=False)
        print("survey-KAT", mode, float(res["pos_intra_loss"]), float(res["pos_inter_loss"]),
              float(res["neg_inter_loss_mean"]), float(res["total"]))
        np.savez_compressed(os.path.join(OUT, f"forward_surveykat_{mode}.npz"), **res)

    # ------------------------------------------------------------------ decay fixtures (A12)
    rows = []
    for kind, cls in (("exp", DD.ExponentialDecay), ("lin", DD.LinearDecay)):
        for init, rate, every, mn in ((0.19, 0.6, 250, 0.0), (11, 0.9, 1000, 5), (0.03, 0.001, 100, 0.01),
                                      (12, 1.0, 300, 0), (1.0, 0.5, 1, 0.1)):
            sched = cls(init, rate, every, mn)
            for step in (0, 1, every - 1, every, 2 * every, 3 * every + 1, 10 * every):
                v = sched.return_update(step)
                rows.append((0 if kind == "exp" else 1, float(init), float(isinstance(init, int)), rate, every,
                             float(mn), step, float(v), float(isinstance(v, int))))
    assert DD.get_depth_scheduler("exp") is DD.ExponentialDecay and DD.get_depth_scheduler("lin") is DD.LinearDecay
    decay = {"table": np.asarray(rows, dtype=np.float64)}

    # live legacy decay: execute the reference's own statements (src/train_segmentation.py:356-375)
    # against a fake `self`, for the four paper_reproduction.sh recipes.
    src_lines = open(os.path.join(REF, "src/train_segmentation.py")).read().split("\n")
    block = "\n".join(l[8:] if l.startswith("        ") else l for l in src_lines[355:375])
    code_obj = compile(block, "<legacy-decay>", "exec")
    recipes = {
        "coco_vits": dict(decay_every_steps=250, depth_feat_shift=0.03, depth_feat_weight=0.19, depth_loss_decay=True,
                          depth_loss_decay_factor=0.6, depth_sampling="fps", fps_sample_decay=True,
                          fps_sample_decay_every_steps=1000, fps_sample_decay_factor=0.9, feature_samples=11, max_steps=7000),
        "coco_vitb": dict(decay_every_steps=300, depth_feat_shift=0.035909146298813595,
                          depth_feat_weight=0.16026274975444096, depth_loss_decay=True, depth_loss_decay_factor=0.64,
                          depth_sampling="fps", fps_sample_decay=True, fps_sample_decay_every_steps=1000,
                          fps_sample_decay_factor=1, feature_samples=12, max_steps=7000),
        "cityscapes": dict(decay_every_steps=400, depth_feat_shift=0.03, depth_feat_weight=0.09, depth_loss_decay=True,
                           depth_loss_decay_factor=0.8, depth_sampling="none", fps_sample_decay=False,
                           fps_sample_decay_every_steps=300, fps_sample_decay_factor=0.9, feature_samples=11, max_steps=7000),
        "potsdam": dict(decay_every_steps=200, depth_feat_shift=0.14, depth_feat_weight=0.13, depth_loss_decay=True,
                        depth_loss_decay_factor=1, depth_sampling="fps", fps_sample_decay=False,
                        fps_sample_decay_every_steps=300, fps_sample_decay_factor=0.9, feature_samples=11, max_steps=7000),
        "fps_until": dict(decay_every_steps=100, depth_feat_shift=0.05, depth_feat_weight=0.2, depth_loss_decay=True,
                          depth_loss_decay_factor=0.5, depth_sampling="fps", fps_sample_decay=True,
                          fps_sample_decay_every_steps=50, fps_sample_decay_factor=0.8, feature_samples=14,
                          fps_until_step=400, post_fps_samples=10, fps_min_samples=6, fix_depth_feat_shift=True,
                          max_steps=600),
    }
    for name, r in recipes.items():
        cfg = SimpleNamespace(fix_depth_feat_shift=False, fps_until_step=0, post_fps_samples=11, fps_min_samples=0)
        for k, v in r.items():
            setattr(cfg, k, v)
        loss_cfg = cfg  # `loss.cfg is model.cfg` in the reference
        fake = SimpleNamespace(cfg=cfg, global_step=0, contrastive_corr_loss_fn=SimpleNamespace(cfg=loss_cfg))
        trace = []
        steps = list(range(0, r["max_steps"], 1))
        for step in steps:
            fake.global_step = step
            exec(code_obj, {"self": fake})
            if step % 50 == 0 or step < 3:
                trace.append((step, cfg.depth_feat_weight, cfg.depth_feat_shift, loss_cfg.feature_samples,
                              0.0 if loss_cfg.depth_sampling == "none" else 1.0))
        decay[f"trace_{name}"] = np.asarray(trace, dtype=np.float64)
        decay[f"recipe_{name}"] = np.asarray(repr(r))
    np.savez_compressed(os.path.join(OUT, "decay.npz"), **decay)
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main()
