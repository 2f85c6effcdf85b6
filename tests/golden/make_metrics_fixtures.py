"""Golden vectors for the validation metrics (UnsupervisedMetrics, src/utils.py:202-319), captured by
IMPORTING the reference on CPU (build container only; import recipe in make_fixtures.py, the torchmetrics stand-in keeps
the states as plain attributes).

    python tests/golden/make_metrics_fixtures.py     # writes tests/golden/metrics.npz
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_fixtures as mf  # noqa: E402


def main():
    mf.import_reference()
    import utils as U  # noqa: E402  (the reference's src/utils.py)
    g = torch.Generator().manual_seed(99)
    fx = {}
    cases = {"e0_hung": (7, 0, True), "e0_plain": (7, 0, False), "e3_hung": (6, 3, True), "c27_hung": (27, 0, True),
             "e5_hung_sparse": (5, 5, True)}
    for name, (n, e, hung) in cases.items():
        m = U.UnsupervisedMetrics("test/cluster/", n, e, hung)
        batches = []
        for b in range(3):
            target = torch.randint(-1, n + 1, (2, 24, 20), generator=g)          # -1 (ignore) and n (out of range) occur
            # predictions correlated with the target through a fixed permutation, some noise, values up to n + e
            perm = torch.randperm(n + e, generator=g)[: n + 1 if e else n]
            noisy = torch.randint(0, n + e + 1, target.shape, generator=g)
            keep = torch.rand(target.shape, generator=g) < 0.7
            mapped = perm[target.clamp(0, len(perm) - 1)]
            preds = torch.where(keep, mapped, noisy)
            if name == "e5_hung_sparse" and b > 0:
                preds = preds.clamp(max=2)                                         # several clusters never predicted
            m.update(preds, target)
            batches.append((preds, target))
        fx[f"{name}_cfg"] = np.asarray([n, e, int(hung)])
        fx[f"{name}_preds"] = np.stack([p.numpy() for p, _ in batches])
        fx[f"{name}_target"] = np.stack([t.numpy() for _, t in batches])
        fx[f"{name}_stats"] = m.stats.numpy()
        out = m.compute()
        fx[f"{name}_miou"] = np.asarray(out["test/cluster/mIoU"])
        fx[f"{name}_acc"] = np.asarray(out["test/cluster/Accuracy"])
        fx[f"{name}_hist"] = np.asarray(m.histogram.numpy(), dtype=np.float64)
        fx[f"{name}_assign0"] = np.asarray(m.assignments[0]).reshape(-1)
        fx[f"{name}_assign1"] = np.asarray(m.assignments[1]).reshape(-1)
        clusters = torch.randint(0, n if (e == 0 or not hung) else n + e, (50,), generator=g)
        fx[f"{name}_clusters"] = clusters.numpy()
        if hung:
            fx[f"{name}_mapped"] = np.asarray(m.map_clusters(clusters))
        # cherry statistics: one more batch through update_cherry / compute_cherry
        m.update_cherry(*batches[0])
        fx[f"{name}_cherry_stats"] = m.cherry_stats.numpy().copy()
        outc = m.compute_cherry()
        fx[f"{name}_cherry_miou"] = np.asarray(outc["test/cluster/mIoU"])
        fx[f"{name}_cherry_acc"] = np.asarray(outc["test/cluster/Accuracy"])
        print(name, out, outc)
    np.savez_compressed(os.path.join(mf.OUT, "metrics.npz"), **fx)


if __name__ == "__main__":
    main()
