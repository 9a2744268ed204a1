"""CPU, gloo, world_size = 2: the data-parallel logic of hidvae_amd.parallel (broadcast, flat-buffer all-reduce, 1/world
scale, per-rank shards and seeds).  The model here is the ORACLE's torch-CPU restatement (the HIP product refuses CPU
tensors by design), so this checks the exchange logic, not the kernels:
  mean over ranks of per-rank gradients == gradient of the concatenated batch, for the untagged step whose losses are
  per-item means (recon.mean + rqvae.mean) -- exactly what DDP gives the reference (train_hidvae.py:709)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import hidvae_amd  # noqa: F401
    from hidvae_amd.parallel import DataParallel, FlatGradBuffer, shard_indices
    from oracle import torch_oracle as O

    cfg = O.Cfg(commitment_weight=0.4, sem_id_uniqueness_weight=0.0)
    # replicas start DIFFERENT on purpose: broadcast must make them identical to rank 0's
    P = O.formula_params(cfg, seed=100 + 7 * rank, with_tags=False)
    params = torch.nn.ParameterDict({k.replace(".", "/"): torch.nn.Parameter(v.clone()) for k, v in P.items()})
    buf = FlatGradBuffer(list(params.parameters()))
    dp = DataParallel(params, buf)
    dp.broadcast_parameters(0)
    ref = O.formula_params(cfg, seed=100, with_tags=False)
    for k, v in ref.items():
        assert torch.equal(params[k.replace(".", "/")].data, v), f"broadcast left {k} different on rank {rank}"

    # every rank owns a shard of the item set and draws its batch from it with a rank-specific seed
    N, B = 64, 16
    X, _, _ = O.formula_batch(cfg, N, seed=9, tagged=False)
    lo, hi = shard_indices(N, rank, world)
    assert (hi - lo) == N // world
    gen = torch.Generator().manual_seed(dp.shard_seed(3))
    idx = lo + torch.randperm(hi - lo, generator=gen)[:B]
    x = X[idx]

    buf.zero()
    Pd = {k.replace("/", "."): p for k, p in params.items()}
    out = O.forward(Pd, cfg, x, training=True)
    out["loss"].backward()
    scale, _ = dp.allreduce()
    assert scale == 1.0 / world
    avg = buf.flat * scale
    torch.save({"idx": idx, "avg": avg.clone(), "seed": dp.shard_seed(3), "order": [k.replace("/", ".") for k in params.keys()]},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_exchange_equals_big_batch(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import torch_oracle as O
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["seed"] != r1["seed"]
    assert set(r0["idx"].tolist()).isdisjoint(set(r1["idx"].tolist())), "ranks drew overlapping items"
    assert torch.equal(r0["avg"], r1["avg"]), "all-reduce left the ranks with different gradients"
    # single-process gradient of the concatenated batch
    cfg = O.Cfg(commitment_weight=0.4, sem_id_uniqueness_weight=0.0)
    P = O.formula_params(cfg, seed=100, with_tags=False)
    X, _, _ = O.formula_batch(cfg, 64, seed=9, tagged=False)
    x = X[torch.cat([r0["idx"], r1["idx"]])]
    _, g = O.grads(P, cfg, x, training=True)
    flat = torch.cat([g[k].reshape(-1) for k in r0["order"]])  # the buffer follows the module's parameter order
    err = float((flat - r0["avg"]).abs().max() / flat.abs().max())
    assert err < 1e-5, err
