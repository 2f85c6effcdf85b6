"""Golden vectors for the segmentation head and the cluster probe, captured by IMPORTING
the reference on CPU (build container only).  DinoFeaturizer cannot be constructed here (its __init__ builds the DINO ViT and
downloads weights), so its `forward` (src/modules.py:90-137) is called unbound on a stand-in object that carries what the method
reads: a backbone `model` whose `get_intermediate_feat` returns seeded tokens, the reference's own `make_clusterer` /
`make_nonlinear_clusterer` heads, `dropout`, `cfg`, `proj_type`.  ClusterLookup (src/modules.py:647-675) imports as is.

    python tests/golden/make_head_fixtures.py     # writes tests/golden/head.npz
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_fixtures as mf  # noqa: E402


class _Tokens(torch.nn.Module):
    """backbone stand-in: get_intermediate_feat(img, n) -> ([tokens (B, 1 + h*w, C)], [attn], [qkv])"""

    def __init__(self, tokens):
        super().__init__()
        self.tokens = tokens

    def get_intermediate_feat(self, img, n=1):
        return [self.tokens], [torch.zeros(1)], [torch.zeros(1)]


def main():
    M, _ = mf.import_reference()
    g = torch.Generator().manual_seed(808)
    fx = {}
    B, C, D, hw, p = 2, 48, 12, 6, 8
    tokens = torch.randn(B, 1 + hw * hw, C, generator=g)
    img = torch.zeros(B, 3, hw * p, hw * p)
    for proj in ("nonlinear", "linear"):
        st = SimpleNamespace(dim=D)
        st.cluster1 = M.DinoFeaturizer.make_clusterer(st, C)
        st.cluster2 = M.DinoFeaturizer.make_nonlinear_clusterer(st, C)
        with torch.no_grad():
            for prm in list(st.cluster1.parameters()) + list(st.cluster2.parameters()):
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.2)
        st.model = _Tokens(tokens)
        st.patch_size, st.feat_type, st.proj_type = p, "feat", proj
        st.cfg = SimpleNamespace(model_type="vit_small", dropout=True)
        st.dropout = torch.nn.Dropout2d(p=.1)
        st.dropout.eval()                                   # eval-mode pass: Dropout2d is the identity -> deterministic
        st.training = False
        feats, code = M.DinoFeaturizer.forward(st, img)
        st.training = True                                  # train-mode arity (masks are random: only shapes are recorded)
        out3 = M.DinoFeaturizer.forward(st, img)
        assert len(out3) == 3
        fx[f"{proj}_feats"] = feats.detach().numpy(); fx[f"{proj}_code"] = code.detach().numpy()
        names = [n for n, _ in list(st.cluster1.named_parameters())] + [n for n, _ in st.cluster2.named_parameters()]
        for i, prm in enumerate(list(st.cluster1.parameters()) + list(st.cluster2.parameters())):
            fx[f"{proj}_w{i}"] = prm.detach().numpy()
        print(proj, tuple(feats.shape), tuple(code.shape), names)
    fx["tokens"] = tokens.numpy()
    # ClusterLookup: hard assignment (alpha None), soft (alpha 2), log_probs
    n_cls = 5
    cl = M.ClusterLookup(D, n_cls)
    with torch.no_grad():
        cl.clusters.copy_(torch.randn(n_cls, D, generator=g))
    x = torch.randn(B, D, hw, hw, generator=g).requires_grad_(True)
    loss_h, probs_h = cl(x, None)
    loss_s, probs_s = cl(x, 2.0)
    logp = cl(x, 2.0, log_probs=True)
    (loss_h + loss_s).backward()
    fx.update(cl_clusters=cl.clusters.detach().numpy(), cl_x=x.detach().numpy(), cl_loss_hard=loss_h.detach().numpy(),
              cl_probs_hard=probs_h.detach().numpy(), cl_loss_soft=loss_s.detach().numpy(), cl_probs_soft=probs_s.detach().numpy(),
              cl_logp=logp.detach().numpy(), cl_grad_x=x.grad.numpy(), cl_grad_clusters=cl.clusters.grad.numpy())
    np.savez_compressed(os.path.join(mf.OUT, "head.npz"), **fx)
    print("wrote head.npz")


if __name__ == "__main__":
    main()
