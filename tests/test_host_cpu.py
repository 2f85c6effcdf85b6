"""CPU-side checks: the C-ABI library loads and exports what include/depthg_corr.h declares, descriptor
validation, the host-side schedules, and that the product path refuses to run without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


def test_library_exports_every_declared_symbol():
    from depthg_amd import _lib
    header = open(os.path.join(ROOT, "include", "depthg_corr.h")).read()
    declared = set(re.findall(r"\b(dg_[a-z_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/depthg_corr.h but not exported"
    assert declared == set(_lib.EXPORTS)
    assert lib.dg_version() == _lib.DG_VERSION == 118


def test_descriptor_validation_and_workspace():
    from depthg_amd import ops
    kw = dict(pointwise=True, zero_clamp=True, stabalize=False, depth_term=True, need_grad=True, shared_coords=False,
              shifts=(0.1, 0.2, 0.3, 0.4), depth_hw=(224, 224))
    small = ops.workspace_bytes(ops.make_desc(2, 64, 70, 14, 14, 11, 5, **kw))
    big = ops.workspace_bytes(ops.make_desc(32, 384, 70, 28, 28, 28, 5, **kw))
    assert 0 < small < big < 2 ** 31
    shared = ops.workspace_bytes(ops.make_desc(32, 384, 70, 28, 28, 28, 5, **{**kw, "shared_coords": True}))
    assert shared < big
    # (C > 768: legal on sample grids of <= 160 positions since round 5 - the fused small-grid kernel streams the channels - and
    #  still refused on larger grids)
    assert ops.workspace_bytes(ops.make_desc(2, 2048, 32, 7, 7, 11, 5, **{**kw, "code_hw": (56, 56)})) > 0
    for bad in (dict(C=769, S=14), dict(C=8193), dict(D=129), dict(n_neg=9), dict(B=0)):
        args = dict(B=2, C=64, D=70, h=14, w=14, S=11, n_neg=5)
        args.update(bad)
        with pytest.raises(RuntimeError):
            ops.workspace_bytes(ops.make_desc(*args.values(), **kw))


def test_struct_layout_matches_header():
    from depthg_amd._lib import CorrDesc
    assert ctypes.sizeof(CorrDesc) == 20 * 4
    assert [f[0] for f in CorrDesc._fields_][-2:] == ["code_h", "code_w"]
    assert [f[0] for f in CorrDesc._fields_][:10] == ["B", "C", "D", "h", "w", "S", "n_neg", "depth_h", "depth_w", "flags"]


def test_product_path_refuses_cpu_tensors():
    from depthg_amd import ContrastiveCorrelationLoss
    from oracle import depthg_oracle as O
    loss = ContrastiveCorrelationLoss(O.default_cfg(feature_samples=4))
    assert list(loss.parameters()) == []          # src/train_segmentation.py:136 iterates .parameters()
    f, c = torch.randn(2, 16, 8, 8), torch.randn(2, 8, 8, 8)
    with pytest.raises(RuntimeError, match="GPU"):
        loss.forward_with(f, f, c, c, torch.ones(2, 1, 16, 16), torch.zeros(2, 4, 4, 2), torch.zeros(2, 4, 4, 2),
                          [torch.zeros(2, dtype=torch.long)] * 5)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "depthg_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in src, f"{fn} mentions the oracle"


def test_decay_schedules_match_reference():
    from depthg_amd import depth_decay as DD
    d = load_golden("decay.npz")
    for kind, init, is_int, rate, every, mn, step, val, val_int in d["table"]:
        cls = DD.get_depth_scheduler("exp" if kind == 0 else "lin")
        init_v, mn_v = (int(init), int(mn)) if is_int else (float(init), float(mn))
        v = cls(init_v, float(rate), int(every), mn_v).return_update(int(step))
        assert float(v) == pytest.approx(float(val), rel=1e-12, abs=1e-15) and isinstance(v, int) == bool(val_int)
    with pytest.raises(NotImplementedError):
        DD.get_depth_scheduler("cos")
    with pytest.raises(AssertionError):
        DD.ExponentialDecay(1.0, 0.5, 10, 0)      # int/float mismatch, like the reference's assert


def test_legacy_decay_traces():
    import ast
    from types import SimpleNamespace
    from depthg_amd.depth_decay import legacy_decay_step
    d = load_golden("decay.npz")
    for key in [k for k in d if k.startswith("trace_")]:
        r = ast.literal_eval(str(d["recipe_" + key[len("trace_"):]]))
        cfg = SimpleNamespace(fix_depth_feat_shift=False, fps_until_step=0, post_fps_samples=11, fps_min_samples=0)
        for k, v in r.items():
            setattr(cfg, k, v)
        want = {int(row[0]): row[1:] for row in d[key]}
        for step in range(int(r["max_steps"])):
            legacy_decay_step(cfg, cfg, step)
            if step in want:
                w = want[step]
                assert cfg.depth_feat_weight == pytest.approx(w[0], rel=1e-12)
                assert cfg.depth_feat_shift == pytest.approx(w[1], rel=1e-12)
                assert cfg.feature_samples == int(w[2]) and (cfg.depth_sampling != "none") == bool(w[3])


def test_super_perm_properties():
    from depthg_amd.loss import super_perm
    torch.manual_seed(0)
    for n in (1, 2, 5, 32):
        for _ in range(20):
            p = super_perm(n, "cpu")
            assert p.shape == (n,) and int(p.min()) >= 0 and int(p.max()) < n
            if n > 1:
                assert not bool((p == torch.arange(n)).any())
    assert super_perm(1, "cpu").tolist() == [0]      # quirk Q6
    from depthg_amd.loss import super_perms
    ps = super_perms(5, 32, "cpu")
    assert ps.shape == (5, 32) and not bool((ps == torch.arange(32)).any()) and int(ps.max()) < 32
    assert all(len(set(row.tolist())) >= 30 for row in ps)          # near-permutations (duplicates possible, Q6)
    assert super_perms(3, 1, "cpu").tolist() == [[0], [0], [0]] and super_perms(0, 4, "cpu").shape == (0, 4)
    g = load_golden("functions.npz")
    torch.manual_seed(3)
    assert np.array_equal(super_perm(8, "cpu").numpy(), g["superperm_8"])   # same RNG consumption as the reference


def test_shard_range_covers_batch():
    from depthg_amd.parallel import shard_range
    for gb, world in ((64, 8), (10, 4), (3, 8)):
        spans = [shard_range(gb, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == gb
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_correspondence_total_matches_fixture_totals():
    """depthg_amd.training.correspondence_total (row A13, src/train_segmentation.py:303-350) on the golden tuples:
    the fixture's `total` was computed by the fixture generator from the reference's own tuple."""
    import torch
    from conftest import FORWARD_CASES, cfg_from_fixture, load_golden
    from depthg_amd.training import correspondence_total, correspondence_weights
    from oracle import depthg_oracle as O
    T = torch.from_numpy
    for case in FORWARD_CASES[:6]:
        fx = load_golden(f"forward_{case}.npz")
        cfg = cfg_from_fixture(fx)
        out = O.forward(cfg, T(fx["feats"]), T(fx["feats_pos"]), T(fx["code"]), T(fx["code_pos"]), T(fx["depth"]),
                        T(fx["depth"]), coords1=T(fx["coords1"]), coords2=T(fx["coords2"]), perms=list(T(fx["perms"])))
        total, logs = correspondence_total(cfg, out)
        assert float(total) == pytest.approx(float(fx["total"]), rel=1e-5, abs=1e-8)
        want_keys = {"loss/pos_intra", "loss/pos_inter", "loss/neg_inter", "cd/pos_intra", "cd/pos_inter", "cd/neg_inter"}
        if cfg.depth_feat_correlation_loss:
            want_keys |= {"loss/depth_feat", "cd/depth_feat"}
        assert set(logs) == want_keys
        assert float(logs["loss/pos_intra"]) == pytest.approx(float(fx["pos_intra_loss"]), rel=1e-5, abs=1e-8)
        # the fused-vector weights give the same total
        depth = bool(cfg.depth_feat_correlation_loss)
        vec = torch.stack([out[0].mean(), out[2].mean(), out[4].mean(), out[6].mean() if depth else torch.zeros(())])
        assert float(torch.dot(vec, correspondence_weights(cfg, depth, "cpu"))) == pytest.approx(float(total), rel=1e-6, abs=1e-9)


def test_correspondence_total_lhp_original_experiment():
    """src/train_segmentation.py:335-337: an experiment name containing "lhp_original" drops the base correspondence term and
    rewrites cfg.lhp_weight to 1.0 (for every later step too); the module picked is the Original class (:82-85)."""
    from types import SimpleNamespace
    from depthg_amd.training import correspondence_total
    cfg = SimpleNamespace(pos_intra_weight=0.5, pos_inter_weight=0.25, neg_inter_weight=2.0, depth_feat_weight=0.1,
                          correspondence_weight=1.0, depth_feat_correlation_loss=True, lhp=True, lhp_weight=0.3,
                          lhp_weight_balance=True, lhp_depth_weight=0.5, experiment_name="run7_lhp_original_a")
    s = lambda v: torch.tensor(float(v))
    out = (s(1), s(0), s(2), s(0), torch.tensor([3.0, 5.0]), s(0), s(6), s(0))
    lhp = (s(10), s(0), s(20), s(0), torch.tensor([30.0, 50.0]), s(0), s(60), s(0))
    total, _ = correspondence_total(cfg, out, lhp)
    assert cfg.lhp_weight == 1.0
    assert abs(float(total) - (0.25 * 20 + 0.5 * 10 + 2.0 * 40 + 0.1 * 0.5 * 60) * 1.0) < 1e-5
    from depthg_amd.segmenter import UnsupervisedSegmenter, default_segmenter_cfg as default_cfg
    from depthg_amd.lhp import OriginalLocalHiddenPositiveProjection, LocalHiddenPositiveProjection
    c2 = default_cfg(lhp=True, experiment_name="x_lhp_original", propagation_strategy="depth", res=64, dino_patch_size=8)
    assert isinstance(UnsupervisedSegmenter(5, c2).lhp_module, OriginalLocalHiddenPositiveProjection)
    c3 = default_cfg(lhp=True, experiment_name="plain", res=64, dino_patch_size=8)
    assert isinstance(UnsupervisedSegmenter(5, c3).lhp_module, LocalHiddenPositiveProjection)


def test_correspondence_total_lhp_and_balance():
    import torch
    from types import SimpleNamespace
    from depthg_amd.training import correspondence_total
    cfg = SimpleNamespace(pos_intra_weight=0.5, pos_inter_weight=0.25, neg_inter_weight=0.75, depth_feat_weight=0.2,
                          correspondence_weight=1.0, depth_feat_correlation_loss=True, lhp=True, lhp_weight=0.3,
                          lhp_weight_balance=True, lhp_depth_weight=0.5)
    s = lambda v: torch.tensor(float(v))
    out = (s(1), s(0), s(2), s(0), torch.tensor([3.0, 5.0]), s(0), s(6), s(0))
    lhp = (s(10), s(0), s(20), s(0), torch.tensor([30.0, 50.0]), s(0), s(60), s(0))
    total, _ = correspondence_total(cfg, out, lhp)
    base = (0.25 * 2 + 0.5 * 1 + 0.75 * 4 + 0.2 * 6) * (1.0 - 0.3)
    extra = (0.25 * 20 + 0.5 * 10 + 0.75 * 40 + 0.2 * 0.5 * 60) * 0.3
    assert float(total) == pytest.approx(base + extra, rel=1e-6)
    with pytest.raises(ValueError):
        correspondence_total(cfg, out)                     # lhp set, second tuple missing
    cfg.depth_feat_correlation_loss = False
    with pytest.raises(ValueError):
        correspondence_total(cfg, out)                     # 8-tuple without the depth term configured


# ---- UnsupervisedMetrics ------------------------------------------------------------------
METRIC_CASES = ["e0_hung", "e0_plain", "e3_hung", "c27_hung", "e5_hung_sparse"]


@pytest.mark.parametrize("case", METRIC_CASES)
def test_metrics_oracle_counts_and_host_compute(case):
    """The oracle's confusion counts equal the reference's stats bit for bit; the host arithmetic of the product class
    (compute / compute_cherry / map_clusters, fed the reference's stats) reproduces the reference's scores."""
    from depthg_amd.metrics import UnsupervisedMetrics
    from oracle import depthg_oracle as O
    g = load_golden("metrics.npz")
    n, e, hung = (int(v) for v in g[f"{case}_cfg"])
    stats = torch.zeros(n + e, n, dtype=torch.int64)
    for p, t in zip(g[f"{case}_preds"], g[f"{case}_target"]):
        stats += O.confusion_counts(torch.from_numpy(p), torch.from_numpy(t), n, e)
    assert np.array_equal(stats.numpy(), g[f"{case}_stats"])
    assert stats[n:].sum() == 0                      # reference quirk: predictions >= n_classes are masked out
    m = UnsupervisedMetrics("test/cluster/", n, e, bool(hung))
    m.stats = stats.clone()
    out = m.compute()
    assert out["test/cluster/mIoU"] == pytest.approx(float(g[f"{case}_miou"]), rel=1e-6)
    assert out["test/cluster/Accuracy"] == pytest.approx(float(g[f"{case}_acc"]), rel=1e-6)
    assert np.array_equal(np.asarray(m.histogram.numpy(), dtype=np.float64), g[f"{case}_hist"])
    assert np.array_equal(np.asarray(m.assignments[0]).reshape(-1), g[f"{case}_assign0"])
    assert np.array_equal(np.asarray(m.assignments[1]).reshape(-1), g[f"{case}_assign1"])
    if hung:
        assert np.array_equal(np.asarray(m.map_clusters(torch.from_numpy(g[f"{case}_clusters"]))), g[f"{case}_mapped"])
    m.cherry_stats = torch.from_numpy(g[f"{case}_cherry_stats"]).clone()
    outc = m.compute_cherry()
    assert outc["test/cluster/mIoU"] == pytest.approx(float(g[f"{case}_cherry_miou"]), rel=1e-6)
    assert outc["test/cluster/Accuracy"] == pytest.approx(float(g[f"{case}_cherry_acc"]), rel=1e-6)
    assert int(m.cherry_stats.sum()) == 0 and m.cherry_stats.device.type == "cpu"


def test_metrics_update_refuses_cpu_tensors():
    from depthg_amd.metrics import UnsupervisedMetrics
    m = UnsupervisedMetrics("x/", 5, 0, True)
    with pytest.raises(RuntimeError, match="GPU"):
        m.update(torch.zeros(4, dtype=torch.long), torch.zeros(4, dtype=torch.long))


# ---- nearest-neighbour table -------------------------------------------------------------
@pytest.mark.parametrize("name", ["small", "wide"])
def test_knn_oracle_matches_reference_calls(name):
    from oracle import depthg_oracle as O
    g = load_golden("knn.npz")
    got = O.knn_table(torch.from_numpy(g[f"{name}_feats"]), 30)
    assert got.dtype == torch.int64 and np.array_equal(got.numpy(), g[f"{name}_nns"])
    assert np.array_equal(got[:, 0].numpy(), np.arange(got.shape[0]))        # column 0 is the image itself


def test_nns_file_format_and_online_pick(tmp_path):
    from depthg_amd import knn
    g = load_golden("knn.npz")
    ref_file = os.path.join(ROOT, "tests", "golden", "nns_fixture.npz")      # written like src/precompute_knns.py:115
    table = knn.load_nns(ref_file, n_images=157)
    assert table.dtype == np.int64 and np.array_equal(table, g["small_nns"])
    path = knn.nns_path(str(tmp_path), "vit_small", "cocostuff27", "train", None, 224)
    assert path.endswith(os.path.join("nns", "nns_vit_small_cocostuff27_train_None_224.npz"))
    assert knn.nns_path("d", "vit_base", "cityscapes", "val", "five", 320).endswith("nns_vit_base_cityscapes_val_five_320.npz")
    knn.save_nns(path, torch.from_numpy(g["small_nns"]))
    with np.load(path) as f:
        assert list(f.keys()) == ["nns"] and f["nns"].dtype == np.int64 and np.array_equal(f["nns"], g["small_nns"])
    with pytest.raises(ValueError, match="precompute_knns"):
        knn.load_nns(os.path.join(str(tmp_path), "missing.npz"))
    with pytest.raises(AssertionError):
        knn.load_nns(path, n_images=3)
    p = load_golden("knn_picks.npz")
    torch.manual_seed(int(p["seed"]))
    picks = [knn.pick_positive(table, ind, int(p["num_neighbors"])) for ind in range(40)]
    assert picks == p["picks"].tolist()
    with pytest.raises(RuntimeError, match="GPU"):
        knn.nearest_neighbors(torch.from_numpy(g["small_feats"]))


def _corr2_asm(tmp_path, src=None, name="dg_corr2.s"):
    import subprocess
    src = src or os.path.join(ROOT, "depthg_amd", "csrc", "dg_corr2.hip")
    out = tmp_path / name
    # (the flags of the Makefile's rule for this file)
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-disable-machine-licm", "--cuda-device-only",
                    "-I", os.path.join(ROOT, "depthg_amd", "csrc"), "-S", str(src), "-o", str(out)], check=True, capture_output=True, timeout=600)
    return out.read_text()


def test_corr2_mfma_results_are_not_touched_early(tmp_path):
    """Every MFMA of dg_corr2.hip is `asm volatile`: hipcc's hazard recogniser does not see them, and round 4's wrong-result bug was
    a compiler-generated v_mov one `s_nop 0` behind one of them.  scripts/mfma_hazards.py walks the generated code along every
    path: no non-MFMA instruction (read OR write) and no MFMA SrcA/SrcB may touch an MFMA's destination registers fewer than 18
    wait states behind it (the kernel's own rule; the matrix pipe's in-order issue is modelled, so "two MFMAs later" is legal).
    The checker must also SEE the round-4 bug: the kernel with scripts/experiments/k_corr2_c5bias_revert.patch applied fails."""
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import mfma_hazards
    bad, stats = mfma_hazards.audit(_corr2_asm(tmp_path))
    assert stats["mfma"] >= 400, stats                      # four instantiations of ~120 MFMAs: the parser found the kernels
    assert not bad, [f"line {J.line_no}: `{J.text}` {t} wait states behind line {M.line_no} `{M.text}`" for M, J, t in bad[:6]]
    # the pre-fix statement of round 4, from the committed patch
    csrc = os.path.join(ROOT, "depthg_amd", "csrc")
    work = tmp_path / "rev"
    work.mkdir()
    shutil.copy(os.path.join(csrc, "dg_corr2.hip"), work / "dg_corr2.hip")
    patch = os.path.join(ROOT, "scripts", "experiments", "k_corr2_c5bias_revert.patch")
    subprocess.run(["patch", "-p3", "-d", str(work), "-i", patch], check=True, capture_output=True)
    bad_rev, _ = mfma_hazards.audit(_corr2_asm(tmp_path, work / "dg_corr2.hip", "dg_corr2_rev.s"))
    assert bad_rev and min(t for _, _, t in bad_rev) <= 4, "the audit no longer sees the round-4 hazard"
    assert any(J.mnem.startswith("v_mov") for _, J, _ in bad_rev)


def test_mfma_hazard_audit_model():
    """The audit's own arithmetic on hand-written snippets: s_nop N = N + 1 wait states, an MFMA in between occupies the matrix pipe
    for its passes, a dependent accumulate (SrcC == vDst) is free, SrcA/SrcB reads and plain writes are not, branches are followed."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from mfma_hazards import audit
    M = "v_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]\n"
    other = "v_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[24:27], a[0:15]\n"
    assert len(audit(M + "s_nop 0\nv_mov_b32 v40, v3\n")[0]) == 1                       # round 4's bug
    assert not audit(M + "s_nop 15\ns_nop 0\nv_mov_b32 v40, v3\n")[0]                 # 1 + 16 + 1 = 18
    assert len(audit(M + "s_nop 15\nv_mov_b32 v40, v3\n")[0]) == 1                      # 17
    assert len(audit(M + M + M + "v_mov_b32 v40, v3\n")[0]) == 1                        # dependent chain: the registers belong to the LAST
    assert not audit(M + M + M + "s_nop 15\ns_nop 0\nv_mov_b32 v40, v3\n")[0]          # MFMA of it, whose own walk decides
    assert not audit(M + other + other + "s_nop 0\nv_mov_b32 v40, v3\n")[0]             # two independent MFMAs later: 8 + 8 + 1 + 1
    assert len(audit(M + other + "v_mov_b32 v40, v3\n")[0]) == 1                         # one is not enough (9)
    assert len(audit(M + "v_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[24:27], a[0:15]\n")[0]) == 1      # result as SrcA of the next MFMA
    assert len(audit(M + "s_nop 3\nv_mov_b32 v2, 0\n")[0]) == 1                          # a WRITE under the late write-back
    assert len(audit(M + "s_cbranch_scc1 .LBB0_2\ns_nop 15\ns_nop 7\n.LBB0_2:\nglobal_store_dword v[30:31], v5, off\n")[0]) == 1
    assert not audit(M + "s_nop 15\ns_nop 7\nv_mov_b32 v40, v3\n", required=18)[0]


def test_inline_asm_arithmetic_outside_corr2_is_audited(tmp_path):
    """hipcc's hazard recogniser does not look inside inline asm.  dg_corr2.hip is covered above; this pins what the OTHER translation
    units may hide from it: no MFMA and no accumulator-register move in an asm statement anywhere else (their MFMAs are builtins:
    the compiler pads them itself), and the one arithmetic asm instruction whose result has a wait-state rule - dg_small.hip's
    `v_dot2c_f32_bf16` (a dot result read by another opcode needs 3 wait states on gfx940+) - passes the audit's dot rule on the
    generated code.  The rule itself is checked on hand-written snippets."""
    import re
    import shutil
    import sys
    csrc = os.path.join(ROOT, "depthg_amd", "csrc")
    asm_stmt = re.compile(r'asm\s*(?:volatile)?\s*\((.*?)\)\s*;', re.S)
    dots = {}
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".h")) or name == "dg_corr2.hip":
            continue
        for m in asm_stmt.finditer(open(os.path.join(csrc, name)).read()):
            body = m.group(1)
            assert "v_mfma" not in body and "v_smfmac" not in body and "v_accvgpr" not in body, (name, body[:80])
            if "v_dot" in body:
                dots[name] = dots.get(name, 0) + 1
    assert set(dots) == {"dg_small.hip"}, dots          # a new asm dot elsewhere: add its file to the audit below
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from mfma_hazards import audit_dots
    D = "v_dot2c_f32_bf16 v5, v9, v9\n"
    assert not audit_dots(D + D + D + "s_nop 1\nv_mul_f32 v7, v5, v5\n")[0]             # 1 + 2 = 3 wait states behind the last one
    assert len(audit_dots(D + D + "s_nop 0\nv_mul_f32 v7, v5, v5\n")[0]) == 1          # 2
    assert len(audit_dots(D + "v_mov_b32 v5, 0\n")[0]) == 1                             # a write counts
    assert not audit_dots(D + "v_mul_f32 v7, v9, v9\n")[0]                              # the sources are free
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    import subprocess
    out = tmp_path / "dg_small.s"
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-I", csrc, "-S",
                    os.path.join(csrc, "dg_small.hip"), "-o", str(out)], check=True, capture_output=True, timeout=600)
    bad, ndot = audit_dots(out.read_text())
    assert ndot >= 8, ndot                                  # the parser found the kernels' norm chains
    assert not bad, [f"line {J.line_no}: `{J.text}` {t} wait states behind line {M.line_no} `{M.text}`" for M, J, t in bad[:6]]


def test_corr2_owns_the_accumulator_file(tmp_path):
    """dg_corr2.hip names accumulator registers literally in inline asm (the stationary feature fragments live there for a
    whole block).  hipcc must neither spill nor allocate values of its own into that file - it does both silently when the
    arch VGPRs run short ("0 spills" in the resource report, v_accvgpr_write / _read around every use).  Audit of the generated
    code: no scratch, no spills, and no accumulator-register instruction outside the kernel's own asm statements."""
    import re
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "depthg_amd", "csrc", "dg_corr2.hip")
    out = tmp_path / "dg_corr2.s"
    # (the flags of the Makefile's rule for this file)
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-disable-machine-licm", "--cuda-device-only", "-S", src,
                    "-o", str(out)], check=True, capture_output=True, timeout=600)
    text = out.read_text()
    assert "scratch_" not in text, "dg_corr2 spills to scratch: its counted vmcnt waits no longer hold"
    inside, stray = False, []
    for line in text.splitlines():
        if ";;#ASMSTART" in line:
            inside = True
        elif ";;#ASMEND" in line:
            inside = False
        elif not inside and re.search(r"\bv_accvgpr_|\ba\[?\d+", line) and not line.lstrip().startswith((";", ".")):
            stray.append(line.strip())
    assert not stray, f"hipcc uses accumulator registers outside the kernel's asm statements: {stray[:5]}"
    m = re.search(r"\.vgpr_spill_count:\s+(\d+)", text)
    assert m and int(m.group(1)) == 0
    # scalar spills land in VGPR lanes (v_writelane / v_readlane): a few in the per-item code are tolerable, none inside a tile loop
    # (blocks of loop depth 2: the kernel walks work items, depth 1, and each item walks its tiles)
    for m in re.finditer(r"\.sgpr_spill_count:\s+(\d+)", text):
        assert int(m.group(1)) <= 8
    depth, in_tile_loop = 0, []
    for line in text.splitlines():
        m = re.search(r"^\.LBB\d+_\d+:.*Depth=(\d+)", line)
        if m:
            depth = int(m.group(1))
        elif re.match(r"^\.LBB\d+_\d+:", line) or re.match(r"^; %bb\.\d+:\s*$", line):
            depth = 0
        elif re.match(r"^; %bb\.\d+:.*Depth=(\d+)", line):
            depth = int(re.search(r"Depth=(\d+)", line).group(1))
        if depth >= 2 and "v_writelane_b32" in line:
            in_tile_loop.append(line.strip())
    assert not in_tile_loop, f"scalar registers spilled inside a tile loop: {in_tile_loop[:4]}"
    m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", text)
    assert m and int(m.group(1)) == 0


@pytest.mark.parametrize("unit", ["dg_post", "dg_prep"])
def test_byte_movers_use_no_scratch_and_no_flat_loads(unit, tmp_path):
    """Two ways hipcc silently made kernels of these files 10 x slower this round, now audited in the generated code:
    (a) a lambda that captures the kernel argument by reference (or is not inlined) puts the whole argument struct, 1.5 KB, into
        SCRATCH in every thread;
    (b) a pointer read back from LDS has no address space: the loads through it become FLAT loads, which count as LDS operations
        too, so every later ds_read waits for all of them.
    No kernel of the operand-preparation / backward-tail files may use scratch, and none may contain a flat load or store."""
    import re
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "depthg_amd", "csrc", unit + ".hip")
    out = tmp_path / (unit + ".s")
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", str(out)],
                   check=True, capture_output=True, timeout=600)
    text = out.read_text()
    flat = [l.strip() for l in text.splitlines() if re.match(r"\s+flat_(load|store)", l)]
    assert not flat, f"{unit}: FLAT memory instructions (a pointer without address space): {flat[:3]}"
    scratch = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)", text)}
    assert scratch, "no kernels found in the generated code"
    bad = {k: v for k, v in scratch.items() if v > 0}
    assert not bad, f"{unit}: kernels with scratch (bytes per thread): {bad}"


def test_bench_cpu_baseline_reports_both_thread_counts(monkeypatch):
    """bench.py's cpu_baseline: the oracle timed at 16 threads and at every host CPU; the second leg runs in a child process
    with a wall-clock limit and is either reported or said to be skipped - it can never hold the bench line up."""
    import os
    import bench
    monkeypatch.setattr(os, "cpu_count", lambda: 24)
    r = bench.cpu_baseline(bench.CONFIGS["C2"], seconds_budget=2.0)
    assert r["kind"] == "port" and r["unit"] == "steps/s" and r["value"] > 0 and r["host_cpus"] == 24
    assert r["cores"] in (16, 24) and "threads" in r["sample"]
    assert len(r["also"]) == 1 and r["also"][0]["cores"] in (16, 24) and r["also"][0]["cores"] != r["cores"]
    assert "value" in r["also"][0] or "skipped" in r["also"][0]


def test_head_draws_only_the_masks_the_reference_draws():
    """DinoFeaturizer calls Dropout2d once per use that exists (src/modules.py:122-132: cluster1's input always,
    cluster2's only for projection_type "nonlinear", the returned feats only with cfg.dropout).  draw_keep_masks draws exactly
    those, in that order, with the generator calls F.dropout2d makes - so the masks AND every later draw of the step (second
    featurizer pass, sample coordinates, super_perm) stay on the reference's random stream in every configuration."""
    import torch
    from depthg_amd.head import draw_keep_masks
    B, C, p = 3, 40, 0.1
    for use in ((True, True, True), (True, False, True), (True, True, False), (True, False, False)):
        torch.manual_seed(11)
        got = draw_keep_masks(B, C, "cpu", p, use=use)
        after = torch.rand(3)
        torch.manual_seed(11)
        drop = torch.nn.Dropout2d(p).train()
        want = [(drop(torch.ones(B, C, 2, 2))[:, :, 0, 0] != 0).float() if u else None for u in use]
        after_ref = torch.rand(3)
        for g, w in zip(got, want):
            assert (g is None and w is None) or torch.equal(g, w)
        assert torch.equal(after, after_ref), use


def test_round4_entry_points_refuse_cpu_tensors_and_bad_state():
    """The entry points added in round 4 have no CPU path either, and the device-generator helpers check their state tensor before a
    pointer reaches the library: forward_pair (two passes whose features do not match), rand_coords_state / keep_masks_state (state
    not the int64[3] device tensor), fps_coords_pair (depth maps that do not match)."""
    from depthg_amd import ops
    from depthg_amd.head import ProjectionHead
    head = ProjectionHead(64, 16).train()
    f = torch.randn(2, 64, 9, 9)
    with pytest.raises(RuntimeError, match="must live on the GPU"):
        head.forward_pair(f, f.clone())
    for bad in (torch.zeros(3, dtype=torch.int64), torch.zeros(4, dtype=torch.int32)):
        with pytest.raises(ValueError, match="state must be"):
            ops.rand_coords_state(bad, (2, 3, 3, 2))
        with pytest.raises(ValueError, match="state must be"):
            ops.keep_masks_state(bad, 4, 64)
    with pytest.raises(RuntimeError, match="must live on the GPU"):
        ops.fps_coords_pair(torch.rand(2, 1, 32, 32), torch.rand(2, 1, 32, 32), (8, 8), 3)


def test_deferred_dropout_host_logic():
    """ops.DeferredDropout and the loss's handling of it, as far as no launch is involved: the wrapper validates its keep flags, the
    loss unwraps to (maps, (keep, keep_pos, scale)), mixed scales materialise the second map, and takes_deferred_dropout follows
    _draw_coords' branch order (salience / depth samplers first, then the dense identity grid)."""
    import torch
    from depthg_amd import ContrastiveCorrelationLoss, ops
    from oracle import depthg_oracle as O
    f, fp = torch.randn(2, 8, 4, 4), torch.randn(2, 8, 4, 4)
    k = (torch.rand(2, 8) > 0.3).float()
    with pytest.raises(ValueError):
        ops.DeferredDropout(f, k[:, :7], 1.1)
    with pytest.raises(ValueError):
        ops.DeferredDropout(f, k, 0.0)
    da = ops.DeferredDropout(f, k, 1.25)
    assert da.shape == f.shape and da.dim() == 4 and torch.equal(da.materialize(), f * (k * 1.25)[:, :, None, None])
    loss = ContrastiveCorrelationLoss(O.default_cfg(feature_samples=4, neg_samples=1, dim=8, dg_dense_grid=True))
    a, b, fk = loss._unwrap_deferred(f, fp)
    assert a is f and b is fp and fk is None
    a, b, fk = loss._unwrap_deferred(da, fp)
    assert a is f and b is fp and torch.equal(fk[0], k) and fk[1] is None and fk[2] == 1.25
    a, b, fk = loss._unwrap_deferred(da, ops.DeferredDropout(fp, k, 2.0))          # two scales: the second map is materialised
    assert a is f and torch.equal(b, fp * (k * 2.0)[:, :, None, None]) and fk[1] is None and fk[2] == 1.25
    assert loss.takes_deferred_dropout((4, 4)) and not loss.takes_deferred_dropout((4, 5)) and not loss.takes_deferred_dropout((4, 4), (8, 8))
    for over in (dict(use_salience=True), dict(depth_sampling="fps"), dict(depth_sampling="simple"), dict(dg_dense_grid=False), dict(feature_samples=3)):
        cfg = O.default_cfg(**{**dict(feature_samples=4, neg_samples=1, dim=8, dg_dense_grid=True), **over})
        assert not ContrastiveCorrelationLoss(cfg).takes_deferred_dropout((4, 4)), over
    # keep flags that do not match the maps are refused before any launch
    with pytest.raises(RuntimeError, match="keep flags"):
        loss.forward_with(f, fp, f, fp, None, torch.zeros(2, 4, 4, 2), torch.zeros(2, 4, 4, 2), None, feat_keep=(k[:1], None, 1.1))


def test_head_refuses_keep_masks_the_kernels_would_misread():
    """The head kernels index keep[b * C + k] through raw pointers: every mask that is not a contiguous fp32 (rows, C) tensor on the
    features' device is refused before a launch (the (B, C) per-pass masks handed to the (2B, C) pair call were read
    out of bounds)."""
    from depthg_amd.head import _check_keeps, ProjectionHead
    B, C = 3, 8
    dev = torch.device("cpu")
    ok = tuple(torch.ones(2 * B, C) for _ in range(3))
    assert _check_keeps(ok, 2 * B, C, dev, "t") == ok
    assert _check_keeps(None, B, C, dev, "t") == (None, None, None)
    assert _check_keeps((ok[0], None, None), 2 * B, C, dev, "t")[1] is None
    bad = {
        "shape (per-pass masks in the pair call)": tuple(torch.ones(B, C) for _ in range(3)),
        "dtype": (torch.ones(2 * B, C, dtype=torch.bool), None, None),
        "strides": (torch.ones(C, 2 * B).t(), None, None),
        "arity": (ok[0], ok[1]),
        "type": (ok[0], 1.0, None),
        "device": (torch.ones(2 * B, C, device="meta"), None, None),
    }
    for what, k in bad.items():
        with pytest.raises(ValueError):
            _check_keeps(k, 2 * B, C, dev, what)
    # the projection_type None path validates too (it multiplies in torch, where a (B,) mask would broadcast silently)
    head = ProjectionHead(C, 4, None).train()
    x = torch.randn(B, C, 2, 2)
    with pytest.raises(ValueError):
        head(x, True, (None, None, torch.ones(C)))
    with pytest.raises(ValueError):
        head.forward_pair(x, x, True, (None, None, torch.ones(B, C)))
    (c, f), (cp, fp) = head.forward_pair(x, x, True, (None, None, torch.ones(2 * B, C)))
    assert torch.allclose(f, x / 0.9)
