#!/usr/bin/env python3
"""Measured parity of the HIP path against the CPU oracle at the headline and at BASELINE.json's configs 2-5, same recipes as
bench.py --config (run on the GPU box):   python scripts/parity_table.py [out.md]
Per configuration: relative error of each loss mean and of the weighted total, relative L2 error and largest element error
(as a fraction of the largest gradient element) of d total / d code and d total / d code_pos.  The table is what the bounds
in tests/test_gpu_parity.py / tests/test_gpu_configs.py are set from (<= 1.5 x the measurement)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from depthg_amd import ContrastiveCorrelationLoss      # noqa: E402
from oracle import depthg_oracle as O                   # noqa: E402

CASES = [("headline", 32, 1234), ("C2", 16, 202), ("C3", 32, 303), ("C4shard", 8, 404), ("C5", 2, 505), ("C5", 8, 506)]


def measure(name, B, seed, dev):
    conf = bench.CONFIGS[name]
    H = dict(conf["H"], B=B)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    f, fp, c, cp, d, dp = bench.synth_inputs(B, seed, "cpu", H)
    g = torch.Generator().manual_seed(seed + 1)
    perms = [O.super_perm(B, g) for _ in range(H["n_neg"])]
    cfg = O.default_cfg(feature_samples=H["S"], neg_samples=H["n_neg"], dim=H["D"], pointwise=conf["pointwise"],
                        depth_sampling=conf["sampling"], dg_outputs="reduced", **conf["scal"])
    if conf["dense"]:
        c1 = c2 = O.identity_coords(B, H["S"])
    elif conf["sampling"] == "fps":
        c1 = O.farthest_point_sampling_depth((H["h"], H["w"]), d, H["S"]) * 2 - 1
        c2 = O.farthest_point_sampling_depth((H["h"], H["w"]), dp, H["S"]) * 2 - 1
    else:
        c1 = torch.rand(B, H["S"], H["S"], 2, generator=g) * 2 - 1
        c2 = torch.rand(B, H["S"], H["S"], 2, generator=g) * 2 - 1
    cr, cpr = c.clone().requires_grad_(True), cp.clone().requires_grad_(True)
    t0 = time.time()
    ref = O.forward(cfg, f, fp, cr, cpr, d, dp, coords1=c1, coords2=c2, perms=perms)
    tot_ref = O.total_loss(cfg, ref)
    tot_ref.backward()
    t_cpu = time.time() - t0
    T = lambda t: t.to(dev)
    cg, cpg = T(c).requires_grad_(True), T(cp).requires_grad_(True)
    out = ContrastiveCorrelationLoss(cfg).forward_with(T(f), T(fp), cg, cpg, T(d), T(c1), T(c2), [T(p) for p in perms],
                                                       shared_coords=conf["dense"], identity_grid=conf["dense"])
    tot = O.total_loss(cfg, out)
    tot.backward()
    torch.cuda.synchronize()
    row = {"config": name, "B": B, "P": H["S"] ** 2, "C": H["C"], "D": H["D"], "oracle_s": round(t_cpu, 2)}
    for i, k in ((0, "intra"), (2, "inter"), (4, "neg"), (6, "depth")):
        a, b = float(out[i].mean()), float(ref[i].mean())
        row[f"rel_{k}"] = abs(a - b) / abs(b)
        row[f"ref_{k}"] = b
    row["rel_total"] = abs(float(tot) - float(tot_ref)) / abs(float(tot_ref))
    row["ref_total"] = float(tot_ref)
    for got, want, k in ((cg.grad.cpu(), cr.grad, "code"), (cpg.grad.cpu(), cpr.grad, "code_pos")):
        row[f"grad_{k}_rel_l2"] = float((got - want).norm() / want.norm())
        row[f"grad_{k}_worst"] = float((got - want).abs().max() / want.abs().max())
    return row


def main():
    dev = torch.device("cuda:0")
    rows = [measure(n, B, s, dev) for n, B, s in CASES]
    lines = ["| config | B | P | C | D | intra | inter | neg | depth | total | grad code rel-L2 / worst | grad code_pos rel-L2 / worst | oracle s |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        e = lambda k: f"{r[k]:.1e}"
        lines.append(f"| {r['config']} | {r['B']} | {r['P']} | {r['C']} | {r['D']} | {e('rel_intra')} | {e('rel_inter')} | {e('rel_neg')} | "
                     f"{e('rel_depth')} | {e('rel_total')} | {e('grad_code_rel_l2')} / {e('grad_code_worst')} | "
                     f"{e('grad_code_pos_rel_l2')} / {e('grad_code_pos_worst')} | {r['oracle_s']} |")
    text = "\n".join(lines)
    print(text)
    print(json.dumps(rows))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            fh.write("# Measured parity, HIP path vs CPU oracle (scripts/parity_table.py; relative errors of the loss means and of the "
                     "weighted total,\n# gradients: relative L2 / largest element error over largest element)\n\n" + text + "\n\n```json\n" + json.dumps(rows, indent=1) + "\n```\n")


if __name__ == "__main__":
    main()
