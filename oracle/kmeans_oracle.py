"""TEST INFRASTRUCTURE -- CPU restatement of the reference's k-means codebook initialisation (init/kmeans.py:34-77).

Only tests/, __graft_entry__.smoke() and bench.py's cpu-baseline legs may import this file; the product never does.
Pinned by tests/golden/kmeans_*.npz (outputs of the reference itself, tests/golden/make_golden.py run_kmeans):
tests/test_oracle_golden.py::test_kmeans_oracle_matches_the_reference.

The reference (per Lloyd iteration): squared distances by broadcasting x[b,1,d] - c[1,k,d] (init/kmeans.py:45-47), argmin with the
first minimum winning (:48), then cluster by cluster the mean of its members, or a re-seed from x when it is empty (:53-60); it stops
when the largest centroid shift is below stop_threshold (:69).  Restated with the same arithmetic per element; the [B, K, D]
difference tensor is built in row chunks to bound memory (the reference builds it whole: 2.6 GB at 20,000 x 1024 x 32)."""
import numpy as np
import torch


def assign(x: torch.Tensor, centroids: torch.Tensor, chunk: int = 2048) -> torch.Tensor:
    out = []
    for i in range(0, x.shape[0], chunk):
        d = ((x[i:i + chunk, None, :] - centroids[None, :, :]) ** 2).sum(dim=2)  # init/kmeans.py:45-47
        out.append(d.min(dim=1).indices)                                         # :48 (first minimum)
    return torch.cat(out)


def lloyd_iteration(x: torch.Tensor, centroids: torch.Tensor, reseed=None):
    """-> (new centroids, assignment).  reseed(cluster) -> row index for an empty cluster (the reference draws torch.randint)."""
    idx = assign(x, centroids)
    new = centroids.clone()
    for c in range(centroids.shape[0]):        # init/kmeans.py:53-60
        m = idx == c
        if not bool(m.any()):
            if reseed is None:
                raise RuntimeError("empty cluster and no reseed source")
            new[c] = x[int(reseed(c))]
        else:
            new[c] = x[m].mean(dim=0)
    return new, idx


def run(x: torch.Tensor, k: int, init_idx, max_iters=None, stop_threshold: float = 1e-10, reseed=None):
    """init/kmeans.py:63-77.  -> (centroids, assignment, iterations run)"""
    x = x.detach().float()
    c = x[torch.as_tensor(np.asarray(init_idx), dtype=torch.int64)].clone()
    a, i = None, 0
    while max_iters is None or i < max_iters:
        old = c
        c, a = lloyd_iteration(x, c, reseed)
        i += 1
        if float(torch.norm(c - old, dim=1).max()) < stop_threshold:
            break
    return c, a, i
