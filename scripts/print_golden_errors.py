#!/usr/bin/env python3
"""developer aid (MI355X): the measured errors of the committed golden / reference-pinned cases, next to the bounds the tests hold
them to - so that a bound can be set from a measurement (tests/test_gpu_parity.py, tests/test_gpu_boundary.py)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as TP
from conftest import load_golden          # noqa
from oracle import depthg_oracle as O
dev = torch.device("cuda:0")
rel = lambda a, b: abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)
for case in TP.FORWARD_CASES:
    fx = load_golden(f"forward_{case}.npz")
    cfg, out, total, g, gp = TP._run_fixture(fx, dev)
    e = {"intra": rel(out[0], fx["pos_intra_loss"]), "inter": rel(out[2], fx["pos_inter_loss"]), "neg": rel(out[4].mean(), fx["neg_inter_loss_mean"]),
         "cd0": rel(out[1].mean(), fx["pos_intra_cd_mean"]), "cd1": rel(out[3].mean(), fx["pos_inter_cd_mean"]), "cd2": rel(out[5].mean(), fx["neg_inter_cd_mean"]),
         "total": rel(total, fx["total"])}
    if cfg.depth_feat_correlation_loss:
        e["depth"] = rel(out[6], fx["depth_feat_loss"])
    gr = []
    for got, want in ((g, fx["grad_code"]), (gp, fx["grad_code_pos"])):
        got = got.cpu().numpy().astype(np.float64); want = want.astype(np.float64)
        gr.append(np.linalg.norm(got - want) / np.linalg.norm(want))
    sub = int(fx["sub"])
    pick = (lambda t: t.detach().cpu().numpy()) if bool(fx["store_full"]) else (lambda t: t.detach().reshape(-1)[::sub].cpu().numpy())
    el = [np.abs(pick(out[1]) - fx["pos_intra_cd"]).max(), np.abs(pick(out[3]) - fx["pos_inter_cd"]).max(), np.abs(pick(out[5]) - fx["neg_inter_cd"]).max(),
          np.abs(pick(out[4]) - fx["neg_inter_loss"]).max()]
    print(f"{case:24s} zero_clamp={int(cfg.zero_clamp)} " + " ".join(f"{k}={v:.1e}" for k, v in e.items()) + f" | grads {gr[0]:.1e} {gr[1]:.1e} | elementwise cd {el[0]:.1e} {el[1]:.1e} {el[2]:.1e} loss {el[3]:.1e}")
