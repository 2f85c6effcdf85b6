"""Host-side mirror of the reference's image-level nearest-neighbour table.

Offline (src/precompute_knns.py:97-115): L2-normalised pooled features `(n_images, F)`, cosine similarities slice by slice
(`n // n_batches` query rows at a time against all images), `topk(30)` indices per row, stored as
`<data_dir>/nns/nns_{model}_{dataset}_{image_set}_{crop}_{res}.npz` under the key `nns` (int64 `(n_images, 30)`; column 0 is
the image itself).  Online (src/data.py:1056-1064, 1079): `ContrastiveSegDataset` loads the table and picks the positive of
image `ind` as `nns[ind][randint(1, num_neighbors + 1)]`.

Here the similarity slice is `dg_knn_similarities` (fp32 MFMA, round 4; a library GEMM before) and the selection is
`dg_topk_rows` (ties broken by the lower index - torch.topk leaves that open); the table, its file name and the online pick are
bit-compatible with the reference.
"""
import os

import numpy as np
import torch

from . import ops

NNS_KEY = "nns"
NNS_K = 30            # src/precompute_knns.py:110


def nns_path(data_dir, model_type, dataset_name, image_set, crop_type, res):
    """File the reference writes (src/precompute_knns.py:72-73) and reads (src/data.py:1056-1057); `crop_type` may be None."""
    return os.path.join(data_dir, "nns", "nns_{}_{}_{}_{}_{}.npz".format(model_type, dataset_name, image_set, crop_type, res))


def nearest_neighbors(normed_feats: torch.Tensor, k: int = NNS_K, n_batches: int = 64) -> torch.Tensor:
    """(n, F) L2-normalised features on the GPU -> int64 (n, k) on the CPU, row i = the k most similar images of i in
    decreasing similarity.  Slices like the reference: `step = n // n_batches` query rows per similarity matrix
    (src/precompute_knns.py:101-112), so the peak scratch is `step x n` floats.  The similarities come from dg_knn_similarities
    (fp32 MFMA, the library's own contraction; 75 TFLOP/s on a 775 x 49,629 x 384 slice): its fp32 dot products run in a fixed k
    order that this library owns, so the table - and with it every positive pick of a training run - has the same bits on every
    ROCm release.  (The vendor GEMM is 15 % faster on the one-shot offline job, 45 against 52 ms for a cocostuff-sized table; it is
    timed beside this path in scripts/knn_time.py and is not a backend of the product.)"""
    if not normed_feats.is_cuda:
        raise RuntimeError(f"depthg_amd: `normed_feats` must live on the GPU (got {normed_feats.device}); there is no CPU path")
    x = normed_feats.detach().to(torch.float32).contiguous()
    n = x.shape[0]
    step = max(n // int(n_batches), 1)
    out = []
    for i in range(0, n, step):
        sims = ops.knn_similarities(x[i:i + step], x)           # einsum("nf,mf->nm"), src/precompute_knns.py:106-108
        out.append(ops.topk_rows(sims, k).cpu())
        del sims
    return torch.cat(out, dim=0)


def save_nns(path, nearest_neighbors_table) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    table = nearest_neighbors_table.cpu().numpy() if isinstance(nearest_neighbors_table, torch.Tensor) else np.asarray(nearest_neighbors_table)
    np.savez_compressed(path, **{NNS_KEY: table.astype(np.int64)})


def load_nns(path, n_images=None) -> np.ndarray:
    if not os.path.exists(path):
        raise ValueError("could not find nn file {} please run precompute_knns".format(path))     # src/data.py:1059-1060
    table = np.load(path)[NNS_KEY]
    if n_images is not None:
        assert n_images == table.shape[0]                                                           # src/data.py:1064
    return table


def pick_positive(nns: np.ndarray, ind: int, num_neighbors: int) -> int:
    """Index of the positive image of `ind` (src/data.py:1079): one of its nearest neighbours 1..num_neighbors, drawn with
    the global torch RNG like the reference."""
    return int(nns[ind][torch.randint(low=1, high=num_neighbors + 1, size=[]).item()])
