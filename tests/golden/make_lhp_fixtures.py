"""Golden vectors for the depth propagation of the LHP branch, captured by IMPORTING the reference on
CPU (build container only).  LocalHiddenPositiveProjection cannot be constructed here (its __init__ calls .cuda()), so
`forward_depth` (src/modules.py:273-339) is called unbound on a stand-in object that only carries `projection_head`:
once with an identity head (the propagated code itself) and once with a seeded 1x1-conv / ReLU / 1x1-conv head.

    python tests/golden/make_lhp_fixtures.py     # writes tests/golden/lhp.npz
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_fixtures as mf  # noqa: E402


def smooth_depth(b, hh, g):
    """A smooth scene (planes + bumps, 8-bit levels) - pooled distances are well separated, like real depth maps."""
    y, x = torch.meshgrid(torch.linspace(0, 1, hh), torch.linspace(0, 1, hh), indexing="ij")
    out = []
    for _ in range(b):
        a, c, e = torch.rand(3, generator=g)
        d = 40 + 120 * (a * x + c * y) + 35 * torch.sin(6.0 * (x + e)) * torch.cos(5.0 * y) + 3 * torch.rand(hh, hh, generator=g)
        out.append(d.clamp(0, 255).round())
    return torch.stack(out).unsqueeze(1)


def main():
    M, _ = mf.import_reference()
    torch.set_num_threads(4)
    g = torch.Generator().manual_seed(404)
    fx = {}
    for name, (b, d, hw, himg) in {"p196": (2, 24, 14, 112), "p784": (1, 16, 28, 224), "rect_pool": (2, 8, 10, 75)}.items():
        code = torch.randn(b, d, hw, hw, generator=g)
        depth = smooth_depth(b, himg, g)
        ident = SimpleNamespace(projection_head=torch.nn.Identity())
        mixed = M.LocalHiddenPositiveProjection.forward_depth(ident, code, depth)
        head = torch.nn.Sequential(torch.nn.Conv2d(d, d, (1, 1)), torch.nn.ReLU(), torch.nn.Conv2d(d, d, (1, 1)))
        with torch.no_grad():
            for prm in head.parameters():
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.3)
        code_g = code.clone().requires_grad_(True)
        proj = M.LocalHiddenPositiveProjection.forward_depth(SimpleNamespace(projection_head=head), code_g, depth)
        up = torch.randn(proj.shape, generator=g)
        (proj * up).sum().backward()
        fx.update({f"{name}_code": code.numpy(), f"{name}_depth": depth.numpy(), f"{name}_mixed": mixed.numpy(),
                   f"{name}_proj": proj.detach().numpy(), f"{name}_up": up.numpy(), f"{name}_grad_code": code_g.grad.numpy()})
        for i, prm in enumerate(head.parameters()):
            fx[f"{name}_head{i}"] = prm.detach().numpy()
        print(name, mixed.shape, float(mixed.abs().max()), float(code_g.grad.abs().max()))
    np.savez_compressed(os.path.join(mf.OUT, "lhp.npz"), **fx)


if __name__ == "__main__":
    main()
