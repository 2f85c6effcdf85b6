"""Golden vectors for the nearest-neighbour table.  src/precompute_knns.py cannot be imported here
(hydra / lightning are absent), so the two torch calls it makes per slice - `einsum("nf,mf->nm")` and `topk(.., 30)[1]`
(src/precompute_knns.py:108-110) - and its slicing (`step = n // n_batches`, :101-104) are executed on seeded inputs, and
its file is written exactly as :115 does; the online pick of src/data.py:1079 is drawn with a seeded global RNG.

    python tests/golden/make_knn_fixtures.py     # writes tests/golden/knn.npz and tests/golden/nns_fixture.npz
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    g = torch.Generator().manual_seed(2024)
    fx = {}
    for name, n, f, nb in (("small", 157, 48, 8), ("wide", 330, 384, 64)):
        feats = F.normalize(torch.randn(n, f, generator=g), dim=1)           # src/precompute_knns.py:19
        step = n // nb
        rows = []
        for i in range(0, n, step):
            sims = torch.einsum("nf,mf->nm", feats[i:i + step, :], feats)
            rows.append(torch.topk(sims, 30)[1])
        nns = torch.cat(rows, dim=0)
        fx[f"{name}_feats"], fx[f"{name}_nns"], fx[f"{name}_nbatches"] = feats.numpy(), nns.numpy(), np.asarray(nb)
        # smallest gap between consecutive similarities inside the top 31 of any row: the fixture is tie-free
        top = torch.topk(torch.einsum("nf,mf->nm", feats, feats), 31)[0]
        fx[f"{name}_min_gap"] = np.asarray(float((top[:, :-1] - top[:, 1:]).min()))
    np.savez_compressed(os.path.join(OUT, "knn.npz"), **fx)
    np.savez_compressed(os.path.join(OUT, "nns_fixture.npz"), nns=fx["small_nns"])        # the reference's file layout (:115)
    torch.manual_seed(5)
    picks = [int(fx["small_nns"][ind][torch.randint(low=1, high=7 + 1, size=[]).item()]) for ind in range(40)]
    np.savez_compressed(os.path.join(OUT, "knn_picks.npz"), picks=np.asarray(picks), seed=np.asarray(5), num_neighbors=np.asarray(7))
    print({k: v.shape if hasattr(v, "shape") else v for k, v in fx.items()}, fx["small_min_gap"], fx["wide_min_gap"])


if __name__ == "__main__":
    main()
