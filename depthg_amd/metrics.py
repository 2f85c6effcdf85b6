"""Host-side mirror of the reference's validation metric.

`UnsupervisedMetrics(prefix, n_classes, extra_clusters, compute_hungarian)` follows src/utils.py:202-319: the confusion
matrix `stats[pred, actual]` (int64, `(n_classes + extra_clusters, n_classes)`) is accumulated on the GPU by
`dg_confusion_update` (the reference: a masked `torch.bincount` per call); `compute()` / `compute_cherry()` / `map_clusters()`
are the reference's host arithmetic (Hungarian matching through scipy, like the reference).  Under data parallelism every
rank accumulates its own shard and `compute()` sums the matrices over the ranks with one all-reduce first - what
torchmetrics does with `dist_reduce_fx="sum"` (src/utils.py:214-220); the local state is left untouched.

Reference behaviour kept on purpose: predictions are masked with `n_classes` (so the extra clusters never reach the
matrix), `compute_cherry` without Hungarian matching reads `stats`, and `map_clusters` inserts the unmatched clusters one
slot late (`missing + 1`).
"""
import numpy as np
import torch
import torch.distributed as dist
from scipy.optimize import linear_sum_assignment

from . import ops


class UnsupervisedMetrics:
    def __init__(self, prefix: str, n_classes: int, extra_clusters: int, compute_hungarian: bool, dist_sync_on_step=True,
                 device=None):
        self.prefix = prefix
        self.n_classes = int(n_classes)
        self.extra_clusters = int(extra_clusters)
        self.compute_hungarian = bool(compute_hungarian)
        self.dist_sync_on_step = dist_sync_on_step
        self.device = torch.device(device) if device is not None else None
        self.reset()

    # -- state -----------------------------------------------------------------------------------------------------
    def _zeros(self, device=None):
        return torch.zeros(self.n_classes + self.extra_clusters, self.n_classes, dtype=torch.int64,
                           device=device if device is not None else (self.device or "cpu"))

    def reset(self):
        self.stats = self._zeros()
        self.cherry_stats = self._zeros()

    def to(self, device):
        self.device = torch.device(device)
        self.stats = self.stats.to(self.device)
        self.cherry_stats = self.cherry_stats.to(self.device)
        return self

    def _accumulate(self, name, preds, target):
        state = getattr(self, name)
        if state.device != preds.device:            # the state follows the first batch's device (it is all-zero or a sum)
            state = state.to(preds.device)
            if self.device is None or name == "stats":
                self.device = preds.device
        ops.confusion_update(state, preds, target, self.n_classes, self.extra_clusters)
        setattr(self, name, state)

    def update(self, preds: torch.Tensor, target: torch.Tensor):          # src/utils.py:222-232
        self._accumulate("stats", preds, target)

    def update_cherry(self, preds: torch.Tensor, target: torch.Tensor):   # src/utils.py:279-289
        self._accumulate("cherry_stats", preds, target)

    def _summed_over_ranks(self, t):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            t = t.clone()
            if dist.get_backend() == "nccl" and not t.is_cuda:      # RCCL reduces device tensors only (an untouched state may still be on the CPU)
                t = t.to(torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    # -- host arithmetic ---------------------------------------------------------------------------------------------
    def map_clusters(self, clusters):                                      # src/utils.py:234-246
        if self.extra_clusters == 0:
            return torch.tensor(self.assignments[1])[clusters]
        unmatched = sorted(set(range(self.n_classes + self.extra_clusters)) - set(self.assignments[0]))
        table = self.assignments[1]
        for c in unmatched:
            table = np.append(table, -1) if c == table.shape[0] else np.insert(table, c + 1, -1)
        return torch.tensor(table)[clusters]

    def _scores(self, stats, plain_stats):
        n, e = self.n_classes, self.extra_clusters
        if self.compute_hungarian:
            cpu = stats.detach().cpu()
            self.assignments = linear_sum_assignment(cpu, maximize=True)
            if e == 0:
                self.histogram = stats[np.argsort(self.assignments[1]), :]
            else:
                self.assignments_t = linear_sum_assignment(cpu.t(), maximize=True)
                rows = stats[self.assignments_t[1], :]
                unmatched = list(set(range(n + e)) - set(self.assignments[0]))
                rest = stats[unmatched, :].sum(0, keepdim=True)
                hist = torch.cat([rows, rest], dim=0)
                self.histogram = torch.cat([hist, torch.zeros(n + 1, 1, device=hist.device)], dim=1)
        else:
            ar = torch.arange(n).unsqueeze(1)
            self.assignments = (ar, ar.clone())
            self.histogram = plain_stats
        tp = torch.diag(self.histogram)
        fp = torch.sum(self.histogram, dim=0) - tp
        fn = torch.sum(self.histogram, dim=1) - tp
        iou = tp / (tp + fp + fn)
        acc = torch.sum(tp) / torch.sum(self.histogram)
        return {self.prefix + "mIoU": 100 * iou[~torch.isnan(iou)].mean().item(), self.prefix + "Accuracy": 100 * acc.item()}

    def compute(self):                                                     # src/utils.py:248-277
        stats = self._summed_over_ranks(self.stats)
        return self._scores(stats, stats)

    def compute_cherry(self):                                              # src/utils.py:291-319
        out = self._scores(self._summed_over_ranks(self.cherry_stats), self._summed_over_ranks(self.stats))
        self.cherry_stats = self._zeros()          # (stays on the metric's device: a CPU state would break an RCCL all-reduce)
        return out
