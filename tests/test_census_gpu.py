"""GPU: near-tie census (SURVEY 7 "hard parts": id flips are REPORTED, not hidden).

Every golden fixture built from formula codebooks has a comfortable top-2 margin; codebooks that k-means produced do not (the survey
measured relative margins down to 1.8e-7 at 65,536 items).  Here the product's own start-up path builds the codebooks -- the HIP
k-means on the residuals of the first 20,000 items, level after level, as train() does -- and then 65,536 items go through the fused
HIP search and through the reference's arithmetic (oracle/torch_oracle.quantize_level: the ATen/MKL expression of
modules/quantize.py:109-113 on the CPU) FROM THE SAME z.  Every disagreement is printed with its float64 margin; the test bounds
how many there may be and requires each one to be a genuine near-tie."""
import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from tests.test_model_gpu import build_model

pytestmark = pytest.mark.gpu
N_ITEMS = 65536
MAX_FLIPS = {256: 12, 1024: 24}      # of N_ITEMS * L decisions (3x256: 196,608; 4x1024: 262,144)
NEAR_TIE_REL = 2e-6                   # a flipped decision's top-2 gap, relative to the distance, must be below this


@pytest.mark.parametrize("L,K", [(3, 256), (4, 1024)])
def test_near_tie_census_on_kmeans_codebooks(L, K):
    from hidvae_amd import _C
    from hidvae_amd.init.kmeans import Kmeans
    cfg = O.Cfg(commitment_weight=0.4, n_layers=L, codebook_size=K, tag_class_counts=[38, 168, 348, 500][:L])
    P = O.formula_params(cfg, seed=100, with_tags=True)
    x, _, _ = O.formula_batch(cfg, N_ITEMS, seed=77, tagged=False)
    m = build_model(cfg, P).eval()
    with torch.no_grad():
        z = m.encode(x.cuda())  # [N,32], L2-normalised (codebook_normalize)
        # start-up as train() does it: level i's codebook = k-means of level i's input residuals over the first 20,000 items
        res = z[:20000].contiguous()
        for i in range(L):
            init = np.random.default_rng(100 + i).choice(res.shape[0], K, replace=False)
            cents = Kmeans(k=K, init_indices=init).run(res).centroids
            m.layers[i].embedding.weight.copy_(cents)
            flags = [j == 0 for j in range(i + 1)]
            cb, cc = _C.codebook_prepare([m.layers[j].embedding.weight.detach() for j in range(i + 1)], flags)
            out = _C.rq_forward(z[:20000].contiguous(), cb, cc, False, _C.MODE_ROTATION, False, 0.4, want_res=True)
            res = out[2][:, i * 32:(i + 1) * 32]  # eval-mode o_i = e_i ...
            res = (out[4][:, i * 32:(i + 1) * 32] - res).contiguous()  # ... so the next level's input is r_i - e_i
        got = m.get_semantic_ids(z, None, None, 0.2).sem_ids.cpu().numpy()
    # the reference's arithmetic on the CPU from the SAME z and the SAME codebooks
    zc = z.cpu()
    Pk = dict(P)
    for i in range(L):
        Pk[f"layers.{i}.embedding.weight"] = m.layers[i].embedding.weight.detach().cpu()
    want = np.empty_like(got)
    r = zc.clone()
    alive = np.ones(N_ITEMS, bool)  # items whose decisions agreed so far (a flip changes every later residual of that item)
    flips = []
    with torch.no_grad():
        for i in range(L):
            cb = O.effective_codebook(Pk, cfg, i)
            o, ids, _, dist = O.quantize_level(r, cb, cfg.codebook_mode, 0.4, False, 0.2, None)
            want[:, i] = ids.numpy()
            bad = np.nonzero(alive & (got[:, i] != want[:, i]))[0]
            for b in bad:
                d64 = ((r[b].double()[None, :] - cb.double()) ** 2).sum(1)
                a, c = int(got[b, i]), int(want[b, i])
                rel = abs(float(d64[a] - d64[c])) / float(d64.min())
                best2 = torch.topk(d64, 2, largest=False).indices.tolist()
                flips.append((int(b), i, a, c, rel, {a, c} == set(best2)))
            alive[bad] = False
            r = r - o
    for b, i, a, c, rel, top2 in flips:
        print(f"[near-tie flip] {L}x{K}: item {b} level {i}: HIP {a} vs torch-CPU {c}, float64 gap / distance = {rel:.2e}, "
              f"the two are the float64 top-2: {top2}")
    print(f"[near-tie census] {L}x{K}: {len(flips)} of {N_ITEMS * L} decisions differ between the HIP search and the ATen/MKL expression "
          f"({N_ITEMS} items, k-means codebooks); {int(alive.sum())} items agree on every level")
    assert len(flips) <= MAX_FLIPS[K], f"{len(flips)} flips"
    for b, i, a, c, rel, top2 in flips:
        assert rel <= NEAR_TIE_REL and top2, (b, i, a, c, rel, top2)
