"""Golden vectors for the attention propagation of LocalHiddenPositiveProjection and for both propagations of
OriginalLocalHiddenPositiveProjection (src/modules.py:235-271, 342-487), captured by IMPORTING the
reference on CPU (build container only).  The reference's constructors call `.cuda()` on their mask tables; for the duration
of this script `torch.Tensor.cuda` is the identity, so the real classes are constructed and their real `index_mask` /
`divide_num` (all zero - see depthg_amd/lhp.py) take part.  Every case is stored once with the constructor's tables and, for
the Original class, once more with `divide_num` set to the neighbourhood sizes (finite outputs and gradients).

    python tests/golden/make_lhp_attn_fixtures.py     # writes tests/golden/lhp_attn.npz
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_fixtures as mf  # noqa: E402
from make_lhp_fixtures import smooth_depth  # noqa: E402


def fake_attention(b, heads, sz, g):
    """Softmax rows over CLS + sz*sz patch tokens: a locality bump plus noise, like a ViT's last block."""
    p = sz * sz
    i, j = torch.arange(p) // sz, torch.arange(p) % sz
    d2 = ((i[:, None] - i[None, :]) ** 2 + (j[:, None] - j[None, :]) ** 2).float()
    logits = 1.5 * torch.randn(b, heads, p + 1, p + 1, generator=g)
    logits[:, :, 1:, 1:] += 3.0 * torch.exp(-d2 / 6.0)
    return torch.softmax(logits, dim=-1)


def seeded_head(module, g):
    with torch.no_grad():
        for prm in module.projection_head.parameters():
            prm.copy_(torch.randn(prm.shape, generator=g) * 0.3)


def run(module, method, code, source, g, fx, name, mixed_only=False):
    """Propagated code with an identity head, then projection + code gradient through the module's own (seeded) head."""
    head = module.projection_head
    module.projection_head = torch.nn.Identity()
    fx[f"{name}_mixed"] = getattr(module, method)(code, source).numpy()
    module.projection_head = head
    print(name, fx[f"{name}_mixed"].shape, "finite" if np.isfinite(fx[f"{name}_mixed"]).all() else "non-finite")
    if mixed_only:
        return
    code_g = code.clone().requires_grad_(True)
    proj = getattr(module, method)(code_g, source)
    up = torch.randn(proj.shape, generator=g)
    (proj * up).sum().backward()
    fx.update({f"{name}_proj": proj.detach().numpy(), f"{name}_up": up.numpy(), f"{name}_grad_code": code_g.grad.numpy()})
    for k, prm in enumerate(head.parameters()):
        fx[f"{name}_head{k}"] = prm.detach().numpy()


def main():
    M, _ = mf.import_reference()
    torch.set_num_threads(4)
    torch.Tensor.cuda = lambda self, *a, **k: self
    g = torch.Generator().manual_seed(909)
    fx = {}
    for name, (b, heads, sz, d) in {"s10": (2, 3, 10, 12), "s12": (1, 2, 12, 66)}.items():
        cfg = SimpleNamespace(dim=d, res=sz * 8, dino_patch_size=8, propagation_strategy="attn")
        code = torch.randn(b, d, sz, sz, generator=g)
        attn = fake_attention(b, heads, sz, g)
        depth = smooth_depth(b, sz * 8, g)
        fx.update({f"{name}_code": code.numpy(), f"{name}_attn": attn.numpy(), f"{name}_depth": depth.numpy()})
        local = M.LocalHiddenPositiveProjection(cfg)
        seeded_head(local, g)
        run(local, "forward_attn", code, attn, g, fx, f"{name}_local_attn")
        orig = M.OriginalLocalHiddenPositiveProjection(cfg)
        seeded_head(orig, g)
        assert int(orig.divide_num.abs().sum()) == 0                  # the constructor's table
        fx[f"{name}_index_mask"] = orig.index_mask.numpy().astype(np.uint8)
        run(orig, "forward_attn", code, attn, g, fx, f"{name}_orig_attn_zero", mixed_only=True)
        run(orig, "forward_depth", code, depth, g, fx, f"{name}_orig_depth_zero", mixed_only=True)
        orig.divide_num = orig.index_mask.float().sum(dim=1, keepdim=True).long()
        fx[f"{name}_counts"] = orig.divide_num.numpy()
        run(orig, "forward_attn", code, attn, g, fx, f"{name}_orig_attn")
        run(orig, "forward_depth", code, depth, g, fx, f"{name}_orig_depth")
    np.savez_compressed(os.path.join(mf.OUT, "lhp_attn.npz"), **fx)


if __name__ == "__main__":
    main()
