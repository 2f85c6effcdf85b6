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
                 "salience", "simple"]   # the last two: coords of the use_salience / depth_sampling='simple' samplers


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


@pytest.fixture(scope="session")
def golden_functions():
    return load_golden("functions.npz")
