"""TEST INFRASTRUCTURE - CPU restatement (torch, fp32) of the reference's segmentation head and probes, the checker for
depthg_amd/head.py's HIP path.  Imported by tests/ only; the product never does.

Pinned against vectors captured from the imported reference classes (tests/golden/head.npz, generator
tests/golden/make_head_fixtures.py; tests/test_oracle_golden.py / tests/test_segmenter.py).

  head_forward         DinoFeaturizer.forward after the backbone, src/modules.py:122-137 with the modules of :75-88
  cluster_lookup       ClusterLookup.forward, src/modules.py:664-675
  probe_cross_entropy  the linear probe's loss, src/train_segmentation.py:427-434
"""
import torch
import torch.nn.functional as F


def conv1x1(x, weight, bias):
    """nn.Conv2d(k, m, (1, 1)): out[b, m, y, x] = sum_k weight[m, k] x[b, k, y, x] + bias[m]"""
    return torch.einsum("mk,bkyx->bmyx", weight.reshape(weight.shape[0], -1), x) + bias.view(1, -1, 1, 1)


def head_forward(feat, w1, b1, w2a=None, b2a=None, w2b=None, b2b=None, keeps=None, p=0.1, feats_dropout=True):
    """code = cluster1(drop1(feat)) [+ cluster2(drop2(feat))], feats = drop3(feat) (src/modules.py:122-132).  `keeps` = three
    (B, C) keep-flag tensors (the Dropout2d noise before its division by 1 - p) or None for eval mode; Dropout2d multiplies whole
    channels by noise / (1 - p)."""
    def drop(i):
        if keeps is None or keeps[i] is None:
            return feat
        noise = keeps[i].to(feat.dtype) / (1.0 - p)
        return feat * noise.view(*noise.shape, 1, 1)
    code = conv1x1(drop(0), w1, b1)
    if w2a is not None:
        hidden = torch.relu(conv1x1(drop(1), w2a, b2a))
        code = code + conv1x1(hidden, w2b, b2b)
    feats = drop(2) if (feats_dropout and keeps is not None) else feat
    return code, feats


def cluster_lookup(x, clusters, alpha, log_probs=False):
    """cosine similarity to the centres; hard (alpha None) or soft assignment; loss = -mean over pixels of sum_n probs * sim"""
    nc = clusters / clusters.norm(dim=1, keepdim=True).clamp_min(1e-12)
    nx = x / x.norm(dim=1, keepdim=True).clamp_min(1e-12)
    sim = torch.einsum("bdyx,nd->bnyx", nx, nc)
    if log_probs:
        return torch.log_softmax(sim * alpha, dim=1)
    if alpha is None:
        probs = torch.zeros_like(sim).scatter_(1, sim.argmax(dim=1, keepdim=True), 1.0)
    else:
        probs = torch.softmax(sim * alpha, dim=1)
    return -(probs * sim).sum(1).mean(), probs


def probe_cross_entropy(logits, label, n_classes):
    """bilinear resize (align_corners=False) of (B, n, h, w) logits to the label resolution, mean cross entropy over the pixels
    whose label is in [0, n_classes)"""
    up = F.interpolate(logits, label.shape[-2:], mode="bilinear", align_corners=False)
    flat = up.permute(0, 2, 3, 1).reshape(-1, logits.shape[1])
    lab = label.reshape(-1)
    ok = (lab >= 0) & (lab < n_classes)
    logp = torch.log_softmax(flat[ok], dim=1)
    return -logp.gather(1, lab[ok].view(-1, 1)).mean()
