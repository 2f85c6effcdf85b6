import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


FORWARD_CASES = ["c1_none", "c1_fps", "nopointwise", "nozeroclamp", "stabalize", "nodepthloss", "zerodepth_fps",
                 "batch1", "S9", "S12", "S14_dim100", "corr_feats", "surveykat_none", "surveykat_fps",
                 "salience", "simple",   # these two: coords of the use_salience / depth_sampling='simple' samplers
                 "fpn_none", "fpn_fps"]  # feature maps (B,C,7,7) next to code maps (B,D,28,28): the FeaturePyramidNet contract


def cfg_from_fixture(fx, **over):
    from oracle import depthg_oracle as O
    kw = {}
    for k in ("feature_samples", "neg_samples"):
        kw[k] = int(fx[k])
    for k in ("pointwise", "zero_clamp", "stabalize", "depth_feat_correlation_loss"):
        kw[k] = bool(fx[k])
    for k in ("pos_intra_shift", "pos_inter_shift", "neg_inter_shift", "depth_feat_shift", "pos_intra_weight",
              "pos_inter_weight", "neg_inter_weight", "depth_feat_weight", "correspondence_weight"):
        kw[k] = float(fx[k])
    kw["depth_sampling"] = str(fx["depth_sampling"])
    if "use_salience" in fx:
        kw["use_salience"] = bool(fx["use_salience"])
    kw.update(over)
    return O.default_cfg(**kw)


def load_golden_seeded(name):
    """Fixtures whose INPUTS are re-drawn from a stored seed instead of being stored (tests/golden/make_round4_fixtures.py:
    the headline-width vectors); the stored checksum pins the draw."""
    import torch
    fx = load_golden(name)
    B, C, D, hf, wf, hc, wc, Himg = (int(v) for v in fx["input_shape"])
    g = torch.Generator().manual_seed(int(fx["input_seed"]))
    f = torch.randn(B, C, hf, wf, generator=g)
    fp = torch.randn(B, C, hf, wf, generator=g)
    c = torch.randn(B, D, hc, wc, generator=g)
    cp = torch.randn(B, D, hc, wc, generator=g)
    d = torch.randint(0, 256, (B, 1, Himg, Himg), generator=g).float()
    dp = torch.randint(0, 256, (B, 1, Himg, Himg), generator=g).float()
    ts = (f, fp, c, cp, d, dp)
    chk = np.asarray([float(t.double().sum()) for t in ts] + [float(t.double().abs().sum()) for t in ts])
    assert np.allclose(chk, fx["input_checksum"], rtol=1e-12, atol=1e-9), "the seeded inputs differ from the ones the reference ran on"
    for k, t in zip(("feats", "feats_pos", "code", "code_pos", "depth", "depth_pos"), ts):
        fx[k] = t.numpy()
    if "coords1" not in fx:          # identity-grid fixture: the pixel-centre grid
        S = int(fx["feature_samples"])
        lin = torch.linspace(-1.0, 1.0, S)
        co = torch.empty(B, S, S, 2)
        co[..., 0] = lin.view(1, 1, S)
        co[..., 1] = lin.view(1, S, 1)
        fx["coords1"] = fx["coords2"] = co.numpy()
    return fx


@pytest.fixture(scope="session")
def golden_functions():
    return load_golden("functions.npz")
