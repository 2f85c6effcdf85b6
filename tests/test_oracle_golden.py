"""Pins the CPU oracle (oracle/depthg_oracle.py) against vectors captured from the reference
itself (tests/golden/*.npz, made by tests/golden/make_fixtures.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import FORWARD_CASES, cfg_from_fixture, load_golden
from oracle import depthg_oracle as O

T = torch.from_numpy


def close(a, b, atol=1e-6, rtol=1e-5):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), atol=atol, rtol=rtol)


def test_norm(golden_functions):
    g = golden_functions
    close(O.norm(T(g["norm_in"])), g["norm_out"], atol=1e-7)


def test_tensor_correlation(golden_functions):
    g = golden_functions
    close(O.tensor_correlation(T(g["corr_a"]), T(g["corr_b"])), g["corr_out"])


def test_sample(golden_functions):
    g = golden_functions
    close(O.sample(T(g["sample_t"]), T(g["sample_coords"])), g["sample_out"], atol=2e-6)


def test_interpolate(golden_functions):
    g = golden_functions
    close(O.interpolate_bilinear_ac(T(g["interp_in"]), (11, 11)), g["interp_out_11"], atol=1e-4, rtol=1e-6)


def test_adaptive_pool_nondivisible(golden_functions):
    g = golden_functions
    out = O.adaptive_avg_pool2d(T(g["fpsd2_depth"]), (14, 14))
    assert np.array_equal(out.numpy(), g["pool2_out"])        # bit exact: the operator's own (sequential, row-major) sum order


@pytest.mark.parametrize("hin,hout", [(224, 28), (448, 56), (100, 14), (64, 9), (300, 14), (37, 5)])
def test_adaptive_pool_is_the_torch_cpu_operator(hin, hout):
    """The pooled depth feeds FPS, whose picks flip on last-bit differences between near-tied distances: the restatement must
    reproduce the operator the reference calls (F.adaptive_avg_pool2d, src/modules.py:1003) bit for bit on general float
    depth, for equal, uneven and large windows."""
    g = torch.Generator().manual_seed(hin + hout)
    d = torch.rand(2, 1, hin, hin, generator=g) * 9 + 0.5
    want = torch.nn.functional.adaptive_avg_pool2d(d, (hout, hout))
    assert np.array_equal(O.adaptive_avg_pool2d(d, (hout, hout)).numpy(), want.numpy())


def test_depth2points_and_factor(golden_functions):
    g = golden_functions
    assert O.fov_factor(90.0).numpy().tobytes() == g["fov_factor"].tobytes()
    out = O.depth2points(T(g["d2p_depth"]), fov=90)
    assert np.array_equal(out.numpy(), g["d2p_out"])  # bit exact: same fp32 op order


def test_fps_indices(golden_functions):
    g = golden_functions
    assert np.array_equal(O.fps(g["fps_points"], 36), g["fps_inds36"])
    assert np.array_equal(O.fps(g["fps_points_flat"], 25), g["fps_inds_flat25"])  # tie-breaking


def test_fps_depth_coords(golden_functions):
    g = golden_functions
    for S in (6, 11):
        c = O.farthest_point_sampling_depth((14, 14), T(g["fpsd_depth"]), S)
        assert np.array_equal(c.numpy(), g[f"fpsd_coords_S{S}"])
    c = O.farthest_point_sampling_depth((14, 14), T(g["fpsd2_depth"]), 5)
    assert np.array_equal(c.numpy(), g["fpsd2_coords_S5"])


def test_super_perm(golden_functions):
    g = golden_functions
    assert np.array_equal(O.super_perm_from(T(g["randperm_8"])).numpy(), g["superperm_8"])
    assert np.array_equal(O.super_perm_from(torch.zeros(1, dtype=torch.long)).numpy(), g["superperm_1"])
    p = g["superperm_8"]
    assert not np.any(p == np.arange(8))


@pytest.mark.parametrize("name,over", [("pw", {}), ("nopw", {"pointwise": False}), ("nozc", {"zero_clamp": False}),
                                       ("stab", {"stabalize": True})])
def test_helper(golden_functions, name, over):
    g = golden_functions
    cfg = O.default_cfg(**over)
    loss, cd = O.helper(cfg, T(g["helper_f1"]), T(g["helper_f2"]), T(g["helper_c1"]), T(g["helper_c2"]), 0.3)
    close(loss, g[f"helper_{name}_loss"], atol=2e-6)
    close(cd, g[f"helper_{name}_cd"], atol=2e-6)


def test_depth_feature_correlation(golden_functions):
    g = golden_functions
    cfg = O.default_cfg()
    c1 = T(g["helper_c1"])
    loss, dd = O.depth_feature_correlation(cfg, c1, c1, T(g["dfc_depth"]), T(g["dfc_depth"]), 0.03)
    close(loss, g["dfc_loss"], atol=2e-6)
    close(dd, g["dfc_dd"], atol=1e-6)
    assert set(np.unique(g["dfc_dd"])) <= {0.0, 1.0}  # quirk Q1


@pytest.mark.parametrize("case", FORWARD_CASES)
def test_forward_against_reference(case):
    fx = load_golden(f"forward_{case}.npz")
    cfg = cfg_from_fixture(fx)
    code = T(fx["code"]).requires_grad_(True)
    code_pos = T(fx["code_pos"]).requires_grad_(True)
    perms = [T(p) for p in fx["perms"]]
    out = O.forward(cfg, T(fx["feats"]), T(fx["feats_pos"]), code, code_pos, T(fx["depth"]), T(fx["depth_pos"]),
                    coords1=T(fx["coords1"]), coords2=T(fx["coords2"]), perms=perms)
    close(out[0], fx["pos_intra_loss"], atol=1e-7, rtol=1e-5)
    close(out[2], fx["pos_inter_loss"], atol=1e-7, rtol=1e-5)
    close(out[4].mean(), fx["neg_inter_loss_mean"], atol=1e-7, rtol=1e-5)
    close(out[1].mean(), fx["pos_intra_cd_mean"], atol=1e-7, rtol=1e-5)
    close(out[5].mean(), fx["neg_inter_cd_mean"], atol=1e-7, rtol=1e-5)
    sub = int(fx["sub"])
    names = [("pos_intra_cd", 1), ("pos_inter_cd", 3), ("neg_inter_loss", 4), ("neg_inter_cd", 5)]
    if cfg.depth_feat_correlation_loss:
        close(out[6], fx["depth_feat_loss"], atol=1e-7, rtol=1e-5)
        names.append(("depth_feat_cd", 7))
    for n, i in names:
        got = out[i] if bool(fx["store_full"]) else out[i].reshape(-1)[::sub]
        close(got, fx[n], atol=3e-6, rtol=1e-5)
    total = O.total_loss(cfg, out)
    close(total, fx["total"], atol=1e-7, rtol=1e-5)
    total.backward()
    gs = max(np.abs(fx["grad_code"]).max(), 1e-12)
    close(code.grad, fx["grad_code"], atol=2e-6 * gs + 1e-9, rtol=1e-4)
    close(code_pos.grad, fx["grad_code_pos"], atol=2e-6 * max(np.abs(fx["grad_code_pos"]).max(), 1e-12) + 1e-9, rtol=1e-4)


@pytest.mark.parametrize("case", ["c1_fps", "zerodepth_fps", "S9", "S12", "fpn_fps"])
def test_forward_fps_coords_regenerated(case):
    """coords drawn by the oracle's own FPS (not injected) equal the reference's."""
    fx = load_golden(f"forward_{case}.npz")
    cfg = cfg_from_fixture(fx)
    c1, c2 = O.draw_coords(cfg, T(fx["feats"]), T(fx["feats_pos"]), T(fx["depth"]), T(fx["depth_pos"]))
    assert np.array_equal(c1.numpy(), fx["coords1"])
    assert np.array_equal(c2.numpy(), fx["coords2"])


@pytest.mark.parametrize("case", ["hl28_rand", "hl28_ident", "fpn2048_none", "fpn2048_fps", "wide1024_ident", "wide1024_rand"])
def test_forward_headline_width_against_reference(case):
    """The headline width (C=384, D=70, 28x28, S=28) at B=2, vectors from the imported reference with its own torch.rand
    coordinates (`rand`) and on the pixel-centre grid (`ident`); inputs re-drawn from the stored seed.  `wide1024_ident` (round 6):
    1024 feature channels on a dense 16 x 16 grid / on 256 sampled positions of 20 x 20 maps (tests/golden/make_round6_fixtures.py)."""
    from conftest import load_golden_seeded
    fx = load_golden_seeded(f"forward_{case}.npz")
    cfg = cfg_from_fixture(fx)
    code = T(fx["code"]).requires_grad_(True)
    code_pos = T(fx["code_pos"]).requires_grad_(True)
    if case.endswith("_fps"):        # (round 5: FeaturePyramidNet's real shapes, (B,2048,7,7) next to (B,32,56,56)) the oracle's own FPS
        c1, c2 = O.draw_coords(cfg, T(fx["feats"]), T(fx["feats_pos"]), T(fx["depth"]), T(fx["depth_pos"]))
        assert np.array_equal(c1.numpy(), fx["coords1"]) and np.array_equal(c2.numpy(), fx["coords2"])
    out = O.forward(cfg, T(fx["feats"]), T(fx["feats_pos"]), code, code_pos, T(fx["depth"]), T(fx["depth_pos"]),
                    coords1=T(fx["coords1"]), coords2=T(fx["coords2"]), perms=[T(p) for p in fx["perms"]])
    for i, k in ((0, "pos_intra_loss"), (2, "pos_inter_loss"), (6, "depth_feat_loss")):
        close(out[i], fx[k], atol=2e-8, rtol=1e-5)
    close(out[4].mean(), fx["neg_inter_loss_mean"], atol=2e-8, rtol=1e-5)
    sub = int(fx["sub"])
    for n, i in (("pos_intra_cd", 1), ("pos_inter_cd", 3), ("neg_inter_loss", 4), ("neg_inter_cd", 5)):
        close(out[i].reshape(-1)[::sub], fx[n], atol=3e-6, rtol=1e-5)
    total = O.total_loss(cfg, out)
    close(total, fx["total"], atol=2e-8, rtol=1e-5)
    total.backward()
    for got, want in ((code.grad, fx["grad_code"]), (code_pos.grad, fx["grad_code_pos"])):
        close(got, want, atol=2e-6 * max(np.abs(want).max(), 1e-12) + 1e-10, rtol=1e-4)


def test_decay_table_and_traces():
    d = load_golden("decay.npz")
    for kind, init, is_int, rate, every, mn, step, val, val_int in d["table"]:
        init_v = int(init) if is_int else float(init)
        mn_v = int(mn) if is_int else float(mn)
        v = O.decay_value("exp" if kind == 0 else "lin", init_v, float(rate), int(every), mn_v, int(step))
        assert float(v) == pytest.approx(float(val), rel=1e-12, abs=1e-15)
        assert isinstance(v, int) == bool(val_int)
    import ast
    for key in d:
        if not key.startswith("trace_"):
            continue
        name = key[len("trace_"):]
        r = ast.literal_eval(str(d[f"recipe_{name}"]))
        cfg = O.default_cfg(**{k: v for k, v in r.items() if k != "max_steps"})
        want = {int(row[0]): row[1:] for row in d[key]}
        for step in range(int(r["max_steps"])):
            O.legacy_decay_step(cfg, cfg, step)
            if step in want:
                w = want[step]
                assert cfg.depth_feat_weight == pytest.approx(w[0], rel=1e-12)
                assert cfg.depth_feat_shift == pytest.approx(w[1], rel=1e-12)
                assert cfg.feature_samples == int(w[2])
                assert (cfg.depth_sampling != "none") == bool(w[3])


# ---- the salience / 'simple' samplers --------------------------------------------------
@pytest.fixture(scope="module")
def golden_samplers():
    return load_golden("samplers.npz")


@pytest.mark.parametrize("seed,S", [(11, 5), (12, 3)])
def test_sample_nonzero_locations(golden_samplers, seed, S):
    """Same RNG calls in the same order as the reference: bit-equal coordinates (incl. the image without non-zeros)."""
    sal = T(golden_samplers["sal_map"])
    torch.manual_seed(seed)
    got = O.sample_nonzero_locations(sal, [sal.shape[0], S, S, 2])
    assert np.array_equal(got.numpy(), golden_samplers[f"sal_coords_seed{seed}_S{S}"])


@pytest.mark.parametrize("name", ["runs", "8bit", "float", "rect"])
def test_simple_depth_informed_sampling(golden_samplers, name):
    g = golden_samplers
    depth, hw, n = T(g[f"simple_{name}_depth"]), tuple(int(v) for v in g[f"simple_{name}_hw"]), int(g[f"simple_{name}_n"])
    assert np.array_equal(O.adaptive_max_pool2d(depth, hw).numpy(), g[f"simple_{name}_pool"])
    torch.manual_seed(int(g[f"simple_{name}_seed"]))
    got = O.simple_depth_informed_sampling(hw, depth, n)
    assert got.shape == (depth.shape[0], n, 1, 2)
    assert np.array_equal(got.numpy(), g[f"simple_{name}_coords"])


@pytest.mark.parametrize("case", ["salience", "simple"])
def test_sampler_coords_regenerated(case):
    """The coordinate draw of the whole forward (src/modules.py:1290-1302), RNG-for-RNG."""
    fx = load_golden(f"forward_{case}.npz")
    cfg = cfg_from_fixture(fx)
    torch.manual_seed(int(fx["rng_seed"]))
    sal = T(fx["salience"]) if "salience" in fx else None
    salp = T(fx["salience_pos"]) if "salience_pos" in fx else None
    c1, c2 = O.draw_coords(cfg, T(fx["feats"]), T(fx["feats_pos"]), T(fx["depth"]), T(fx["depth_pos"]), sal, salp)
    assert np.array_equal(c1.numpy(), fx["coords1"])
    assert np.array_equal(c2.numpy(), fx["coords2"])


def test_uniform_rank_samplers_consistent():
    """The uniform-driven variants (what the HIP samplers implement) pick, for uniforms that map to the same ranks, the
    same pixels as the RNG-driven ones; and every picked pixel is a legal candidate."""
    g = torch.Generator().manual_seed(5)
    sal = (torch.rand(2, 9, 7, generator=g) > 0.6).float()
    u = torch.rand(2, 16, generator=g).numpy()
    ufb = torch.rand(2, 16, 2, generator=g).numpy()
    c = O.sample_nonzero_locations_from_uniform(sal, [2, 4, 4, 2], u, ufb)
    xy = ((c + 1) / 2 * sal.shape[1]).round().long()                 # flipped: (col, row), both scaled by H
    for b in range(2):
        for s in range(16):
            x, y = xy[b].reshape(-1, 2)[s]
            assert sal[b, y, x] != 0
    # rank 0 / last rank
    first = O.sample_nonzero_locations_from_uniform(sal, [2, 1, 1, 2], np.zeros((2, 1), np.float32), ufb[:, :1])
    last = O.sample_nonzero_locations_from_uniform(sal, [2, 1, 1, 2], np.full((2, 1), 0.99999994, np.float32), ufb[:, :1])
    nz = torch.nonzero(sal[0])
    assert tuple(((first[0, 0, 0] + 1) / 2 * 9).round().long().tolist()) == (int(nz[0, 1]), int(nz[0, 0]))
    assert tuple(((last[0, 0, 0] + 1) / 2 * 9).round().long().tolist()) == (int(nz[-1, 1]), int(nz[-1, 0]))
    depth = torch.randint(0, 3, (2, 1, 20, 20), generator=g).float()
    uv, up = torch.rand(2, 12, generator=g).numpy(), torch.rand(2, 12, generator=g).numpy()
    cs = O.simple_depth_informed_sampling_from_uniform((5, 5), depth, 12, uv, up)
    assert cs.shape == (2, 12, 1, 2)
    rc = (cs * 5 - 0.5).round().long()
    pooled = (O.adaptive_max_pool2d(depth, (5, 5)) * 10).round() / 10
    srt = [np.sort(pooled[b].reshape(-1).numpy()) for b in range(2)]
    for b in range(2):
        for s in range(12):
            r, cc = rc[b, s, 0]
            assert float(pooled[b, 0, r, cc]) == float(srt[b][O.rank_from_uniform(uv[b, s], 25)])


# ---- depth propagation of the LHP branch -------------------------------------------------
@pytest.mark.parametrize("name", ["p196", "p784", "rect_pool"])
def test_lhp_propagation_against_reference(name):
    """The oracle uses the direct distance formula; the reference's torch.cdist takes its matmul path for P > 25 (non-zero
    self-distance, about 1e-7 * |point|^2 noise on d^2), so agreement is 5e-4 relative, not bitwise - and exact where the
    neighbour sets are the same and cdist is exact (the 10x10 case)."""
    g = load_golden("lhp.npz")
    out = O.lhp_propagate(T(g[f"{name}_code"]), T(g[f"{name}_depth"]))
    ref = g[f"{name}_mixed"]
    rel = np.linalg.norm(out.numpy() - ref) / np.linalg.norm(ref)
    assert rel < 5e-4, rel
    wmap, stats = O.lhp_depth_weights(T(g[f"{name}_depth"]), g[f"{name}_code"].shape[-2:])
    kept = (wmap > 0).sum(-1)
    p = wmap.shape[-1]
    assert int(kept.min()) >= int(np.floor(0.01 * (p - 1))) + 1            # at least the quantile's lower rank + 1 survive
    assert torch.all(torch.diagonal(wmap, dim1=1, dim2=2) == 1.0)          # self-distance 0 -> weight 1
    assert torch.all(stats[..., 0] == 0.0)


# ---- N3, the other propagation maps: attention strategy and the Original class (tests/golden/lhp_attn.npz) -------------
def _head_from(g, key, d):
    head = torch.nn.Sequential(torch.nn.Conv2d(d, d, (1, 1)), torch.nn.ReLU(), torch.nn.Conv2d(d, d, (1, 1)))
    with torch.no_grad():
        for k, prm in enumerate(head.parameters()):
            prm.copy_(T(g[f"{key}_head{k}"]))
    return head


@pytest.mark.parametrize("name", ["s10", "s12"])
def test_lhp_attention_map_against_reference(name):
    """forward_attn of LocalHiddenPositiveProjection (src/modules.py:235-271): same thresholded map (the oracle's ordered heads
    sum and quantile are the reference's), weighted mean to float32 summation-order accuracy, gradient through the head."""
    g = load_golden("lhp_attn.npz")
    code, attn = T(g[f"{name}_code"]), T(g[f"{name}_attn"])
    key = f"{name}_local_attn"
    out = O.lhp_propagate_attn(code, attn)
    np.testing.assert_allclose(out.numpy(), g[f"{key}_mixed"], rtol=2e-5, atol=1e-6)
    wmap = O.lhp_attn_weights(attn)
    p = wmap.shape[-1]
    zeroed = (wmap == 0).sum(-1)
    # the row minimum (normalised to 0) and everything above the 99 % quantile: P - 1 - floor(0.99 (P - 1)) values
    assert int(zeroed.min()) >= 1 and int(zeroed.max()) <= p - int(np.floor(np.float32(0.99) * np.float32(p - 1))) + 1
    cg = code.clone().requires_grad_(True)
    proj = _head_from(g, key, code.shape[1])(torch.einsum("bpq,bdq->bdp", wmap, cg.reshape(*cg.shape[:2], -1)).reshape(cg.shape) / p)
    np.testing.assert_allclose(proj.detach().numpy(), g[f"{key}_proj"], rtol=1e-4, atol=1e-5)
    (proj * T(g[f"{key}_up"])).sum().backward()
    np.testing.assert_allclose(cg.grad.numpy(), g[f"{key}_grad_code"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", ["s10", "s12"])
@pytest.mark.parametrize("source", ["attn", "depth"])
def test_lhp_original_variant_against_reference(name, source):
    """OriginalLocalHiddenPositiveProjection (src/modules.py:342-487): the index mask is the clipped 3x3 neighbourhood; with the
    constructor's all-zero divide_num the output is sum / 0 (same inf / nan pattern); with the neighbourhood sizes it is finite
    and matches to float32 summation-order accuracy (depth: the cdist tolerance of the depth strategy's test)."""
    g = load_golden("lhp_attn.npz")
    code = T(g[f"{name}_code"])
    sz = code.shape[-1]
    assert np.array_equal(O.lhp_index_mask(sz), g[f"{name}_index_mask"].astype(np.float32))
    counts = T(g[f"{name}_counts"])
    assert np.array_equal(O.lhp_index_mask(sz).sum(1), counts.numpy()[:, 0])
    wmap = (O.lhp_original_attn_weights(T(g[f"{name}_attn"]), sz) if source == "attn"
            else O.lhp_original_depth_weights(T(g[f"{name}_depth"]), sz))
    zero = O.lhp_original_propagate(wmap, code, torch.zeros(sz * sz, 1, dtype=torch.long)).numpy()
    ref0 = g[f"{name}_orig_{source}_zero_mixed"]
    assert not np.isfinite(zero).any() and not np.isfinite(ref0).any()
    same = (np.isnan(zero) == np.isnan(ref0)) & (np.isnan(zero) | (np.sign(zero) == np.sign(ref0)))
    assert same.mean() > 0.999, same.mean()                                   # (a sum that rounds across 0 may flip one sign)
    key = f"{name}_orig_{source}"
    out = O.lhp_original_propagate(wmap, code, counts)
    tol = dict(rtol=2e-5, atol=2e-6) if source == "attn" else dict(rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(out.numpy(), g[f"{key}_mixed"], **tol)
    cg = code.clone().requires_grad_(True)
    proj = _head_from(g, key, code.shape[1])(O.lhp_original_propagate(wmap, cg, counts))
    (proj * T(g[f"{key}_up"])).sum().backward()
    np.testing.assert_allclose(cg.grad.numpy(), g[f"{key}_grad_code"], rtol=tol["rtol"] * 5, atol=tol["atol"] * 5)
