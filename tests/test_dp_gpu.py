"""GPU: the PRODUCT's data-parallel step (hidvae_amd.step.GraphedTrainStep with hidvae_amd.parallel.DataParallel).

* one rank, RCCL: the overlapped exchange -- captured INSIDE the step's one graph (graph[fwd, backward part 1, all-reduce(bucket 1) ||
  backward part 2, all-reduce(bucket 2), AdamW], the default over RCCL) and in the three-graph form host-side backends get --
  must equal the plain single-GPU step bit for bit: the split backward, the bucket-first flat gradient buffer and the in-place
  gradient writes change no arithmetic;
* two ranks sharing this box's one GPU (backend gloo, which moves device tensors through the host; RCCL refuses two ranks on one
  device): after several graphed steps on different per-rank batches the replicas are bit-identical, and equal to ONE process
  stepping on the concatenated batch (mean of per-rank gradients == gradient of the per-item-mean loss over the union)."""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from tests.test_model_gpu import build_model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(commitment_weight=0.4, tag_alignment_weight=0.15, tag_prediction_weight=0.55, tag_class_counts=[38, 168, 348], use_focal_loss=True,
           focal_loss_params={"gamma_0": 2.7, "alpha_0": 0.24}, dropout_rate=0.4, sem_id_uniqueness_weight=1.5, sem_id_uniqueness_margin=0.0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _groups(m, tagged):
    g = [{"params": list(m.encoder.parameters()) + list(m.decoder.parameters()), "lr": 2.8e-4, "weight_decay": 0.015},
         {"params": [p for layer in m.layers for p in layer.parameters()], "lr": 2.8e-4, "weight_decay": 0.015}]
    if tagged:
        for i in range(m.n_layers):
            g.append({"params": list(m.tag_predictors[i].parameters()), "lr": 2.8e-4 * (1 + 0.1 * i), "weight_decay": 0.015 / (1 + 0.2 * i)})
            g.append({"params": list(m.tag_projectors[i].parameters()), "lr": 2.8e-4 * (1 + 0.1 * i), "weight_decay": 0.015 / (1 + 0.2 * i)})
    return g


def _batch(cfg, B, seed, tagged):
    x, te, ti = O.formula_batch(cfg, B, seed=seed, tagged=tagged)
    b = types.SimpleNamespace(x=x.cuda())
    if tagged:
        b.tags_emb, b.tags_indices = te.cuda(), ti.cuda()
    return b


def _run(tagged, dp_mode, steps=6, B=128, in_graph=True, fail_first_capture=False):
    """dp_mode None: plain step; 'overlap': DataParallel over a one-rank RCCL group, collectives issued anyway.  in_graph: the
    collectives are captured inside the step's ONE graph (the default over RCCL); False: kept between three graphs (what host-side
    backends get)"""
    from hidvae_amd.optim import HidvaeAdamW
    from hidvae_amd.parallel import DataParallel
    from hidvae_amd.step import GraphedTrainStep
    cfg = O.Cfg(**CFG)
    m = build_model(cfg, O.formula_params(cfg, seed=100, with_tags=True)).train()
    torch.manual_seed(1234)
    multi = dp_mode is not None
    opt = HidvaeAdamW(_groups(m, tagged), cosine=(1000, 7e-8), flat_grads=multi, first_bucket=m.dp_first_bucket(B) if multi else None).prepare()
    dp = None
    if multi:
        dp = DataParallel(m, opt.grad_buffer)
        dp.always = True
    st = GraphedTrainStep(m, opt, [_batch(cfg, B, 500, tagged)], dp=dp, gumbel_t=0.2, warmup=2, overlap=True if multi else None)
    if multi and not in_graph:
        st._collectives_capturable = lambda: False
    if fail_first_capture:
        # a communicator that refuses to be captured: the first collective issued under capture raises, once
        real, state = dp.allreduce_part, {"raised": 0}

        def refusing(lo, hi):
            if torch.cuda.is_current_stream_capturing() and not state["raised"]:
                state["raised"] += 1
                raise RuntimeError("stand-in: this communicator cannot be captured")
            return real(lo, hi)

        dp.allreduce_part = refusing
    rows = []
    for it in range(steps):
        rows.append(st([_batch(cfg, B, 500 + it, tagged)]).clone())
    if multi:  # the exchange ran on the package's own communicator (hidvae_amd.rccl), not on the process group's
        from hidvae_amd.rccl import Communicator
        assert isinstance(dp.communicator(), Communicator) and dp.capturable()
    between = multi and (not in_graph or fail_first_capture)
    assert st.graphs is not None and len(st.graphs) == (3 if between else 1)
    assert st.in_graph == (multi and not between)
    if fail_first_capture:
        assert state["raised"] == 1 and st._in_graph_failed
    return torch.stack(rows).cpu(), {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


def _one_rank_rccl_worker(rank, port, tagged, in_graph, fail_first_capture=False):
    """body of test_overlapped_dp_step_equals_the_plain_step_bit_for_bit, in a process of its own: a process group (RCCL: watchdog and
    proxy threads) lives and dies with it instead of being created and destroyed inside the pytest process"""
    sys.path.insert(0, ROOT)
    import hidvae_amd  # noqa: F401  (before the first HIP call: the graph-queue setting is read when the runtime initialises)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))  # RCCL, one rank: the --dist 1 rehearsal as a test
    try:
        rows1, sd1 = _run(tagged, "overlap", in_graph=in_graph, fail_first_capture=fail_first_capture)
        rows0, sd0 = _run(tagged, None)
    finally:
        dist.destroy_process_group()
    assert torch.equal(rows1, rows0), "logged loss rows differ between the overlapped DP step and the plain step"
    for k in sd0:
        assert torch.equal(sd1[k], sd0[k]), k


@pytest.mark.timeout(600)
@pytest.mark.parametrize("in_graph", [True, False], ids=["collectives_in_the_graph", "collectives_between_graphs"])
@pytest.mark.parametrize("tagged", [False, True], ids=["untagged", "tagged"])
def test_overlapped_dp_step_equals_the_plain_step_bit_for_bit(tagged, in_graph):
    import torch.multiprocessing as mp
    mp.spawn(_one_rank_rccl_worker, args=(_free_port(), tagged, in_graph), nprocs=1, join=True)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("tagged", [False, True], ids=["untagged", "tagged"])
def test_a_refused_in_graph_capture_falls_back_to_collectives_between_graphs(tagged):
    """GraphedTrainStep._capture's fallback: the capture with the collectives inside raises part-way (host bookkeeping of a step that
    never ran is left behind: optimizer `_prepared` / `_early`, sealed flat buffer, first-bucket marks, forked level streams); the
    step must reset that, re-capture in the between-graphs form and still equal the plain step bit for bit"""
    import torch.multiprocessing as mp
    mp.spawn(_one_rank_rccl_worker, args=(_free_port(), tagged, True, True), nprocs=1, join=True)


def _event_query_worker(rank):
    """What aborted round 3's first captured data-parallel step, in isolation: a completion-event query from ANOTHER thread (the process
    group's watchdog does this every 100 ms for collectives it has not yet seen complete) while this thread captures."""
    import threading
    torch.cuda.set_device(0)
    x = torch.ones(1024, device="cuda")
    s = torch.cuda.Stream()
    ev = torch.cuda.Event()
    with torch.cuda.stream(s):
        y = x * 2
        ev.record(s)
    torch.cuda.synchronize()

    def capture(mode):
        seen = {}

        def poll():
            try:
                seen["done"] = ev.query()
            except Exception as e:  # noqa: BLE001
                seen["error"] = f"{type(e).__name__}: {str(e).splitlines()[0]}"

        g = torch.cuda.CUDAGraph()
        failed = None
        try:
            with torch.cuda.graph(g, capture_error_mode=mode):
                z = y + 1
                t = threading.Thread(target=poll)
                t.start()
                t.join()
                z = z * 3
        except RuntimeError as e:
            failed = str(e).splitlines()[0]
        return seen, failed, g, (None if failed else z)

    # thread-local capture: the other thread's query is legal, the capture completes and replays
    seen, failed, g, z = capture("thread_local")
    assert seen == {"done": True} and failed is None, (seen, failed)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(z, torch.full_like(z, 9.0))
    # global capture (torch's default): the same query is an error -- in the watchdog's C++ thread that error is an uncaught exception
    seen, failed, g, z = capture("global")
    assert "error" in seen or failed is not None, ("a query from another thread did not disturb a global-mode capture", seen, failed)
    print(f"[event-query] global-mode capture: query -> {seen}; capture -> {failed}", flush=True)


@pytest.mark.timeout(300)
def test_event_query_from_another_thread_breaks_a_global_capture_and_not_a_thread_local_one():
    """the mechanism behind GraphedTrainStep._capture_mode (no process group involved: deterministic, one run)"""
    import torch.multiprocessing as mp
    mp.spawn(_event_query_worker, nprocs=1, join=True)


def _joined_stream_query_worker(rank):
    """Why the process group's collectives are never captured (hidvae_amd/rccl.py): an event recorded EAGERLY on a stream, long
    complete, cannot be queried any more once that stream has joined a capture -- whatever the capture mode, whichever thread asks.
    ProcessGroupNCCL's internal stream joins the capture with the first captured collective; its watchdog's poll of an earlier eager
    collective is exactly this query."""
    import threading
    torch.cuda.set_device(0)
    x = torch.zeros(1 << 16, device="cuda")
    s, c = torch.cuda.Stream(), torch.cuda.Stream()
    ev = torch.cuda.Event()
    with torch.cuda.stream(s):
        x.add_(1.0)
        ev.record(s)  # an "eager collective" of a warm-up step on the group's internal stream
    torch.cuda.synchronize()
    assert ev.query()
    seen = {}

    def poll(tag):
        try:
            seen[tag] = ev.query()
        except Exception as e:  # noqa: BLE001
            seen[tag] = f"{type(e).__name__}: {str(e).splitlines()[0]}"

    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(c), torch.cuda.graph(g, stream=c, capture_error_mode="thread_local"):
            x.mul_(2.0)
            t = threading.Thread(target=poll, args=("before the stream joins",))
            t.start(), t.join()
            fork = torch.cuda.Event()
            fork.record(c)
            s.wait_event(fork)  # the first captured collective: the internal stream joins the capture
            with torch.cuda.stream(s):
                x.add_(3.0)
            t = threading.Thread(target=poll, args=("after the stream joined",))
            t.start(), t.join()
            c.wait_stream(s)
    except RuntimeError as e:  # (the refused query may also leave the capture unable to end: not what is tested here)
        seen["capture"] = str(e).splitlines()[0]
    print(f"[joined-stream query] {seen}", flush=True)
    assert seen["before the stream joins"] is True, seen
    assert isinstance(seen["after the stream joined"], str) and "captur" in seen["after the stream joined"], seen


@pytest.mark.timeout(300)
def test_event_of_a_stream_that_joined_a_capture_cannot_be_queried():
    """the second way a process group's watchdog kills a captured data-parallel step (deterministic, no process group involved)"""
    import torch.multiprocessing as mp
    mp.spawn(_joined_stream_query_worker, nprocs=1, join=True)


def _rank_worker(rank, world, port, out_dir, steps, B, tagged=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import hidvae_amd  # noqa: F401  (before the first HIP call: the graph-queue setting is read when the runtime initialises)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from hidvae_amd.optim import HidvaeAdamW
    from hidvae_amd.parallel import DataParallel
    from hidvae_amd.step import GraphedTrainStep
    cfg = O.Cfg(**CFG)
    # replicas start DIFFERENT on purpose: broadcast_parameters must make them rank 0's
    m = build_model(cfg, O.formula_params(cfg, seed=100 + 13 * rank, with_tags=True)).train()
    opt = HidvaeAdamW(_groups(m, tagged), cosine=(1000, 7e-8), flat_grads=True, first_bucket=m.dp_first_bucket(B)).prepare()
    dp = DataParallel(m, opt.grad_buffer)
    dp.broadcast_parameters(0)
    st = GraphedTrainStep(m, opt, [_batch(cfg, B, 700, tagged)], dp=dp, gumbel_t=0.2, warmup=2, overlap=True)
    for it in range(steps):
        st([_batch(cfg, B, 700 + 2 * it + rank, tagged)])
    torch.cuda.synchronize()
    assert st.graphs is not None and len(st.graphs) == 3
    # (BatchNorm running statistics stay rank-local, as under DDP: each rank saw its own batches)
    keep = (lambda k: "running_" not in k and "num_batches" not in k) if tagged else (lambda k: not k.startswith("tag_"))
    torch.save({k: v.detach().cpu() for k, v in m.state_dict().items() if keep(k)}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_tagged_ranks_stay_bit_identical(tmp_path):
    """The TAGGED overlapped step on two ranks (gloo, both on this GPU): tag streams + the heads' early backward + the split backward +
    two exchanges between three graphs -- replicas that start different must end bit-identical in every parameter, tag heads included,
    and must have moved."""
    import torch.multiprocessing as mp
    world, steps, B = 2, 4, 64
    mp.spawn(_rank_worker, args=(world, _free_port(), str(tmp_path), steps, B, True), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert any(k.startswith("tag_predictors") for k in r0) and any(k.startswith("tag_projectors") for k in r0)
    for k in r0:
        assert torch.equal(r0[k], r1[k]), f"replicas diverged: {k}"
        assert torch.isfinite(r0[k]).all(), k
    cfg = O.Cfg(**CFG)
    start = O.formula_params(cfg, seed=100, with_tags=True)
    moved = [k for k in r0 if k in start and r0[k].dtype.is_floating_point and not torch.equal(r0[k], start[k])]
    assert len(moved) > 0.9 * sum(1 for k in r0 if k in start and r0[k].dtype.is_floating_point)


@pytest.mark.timeout(600)
def test_two_ranks_stay_bit_identical_and_match_the_big_batch(tmp_path):
    import torch.multiprocessing as mp
    from hidvae_amd.optim import HidvaeAdamW
    from hidvae_amd.step import GraphedTrainStep
    world, steps, B = 2, 5, 96
    mp.spawn(_rank_worker, args=(world, _free_port(), str(tmp_path), steps, B), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    for k in r0:
        assert torch.equal(r0[k], r1[k]), f"replicas diverged: {k}"
    # ONE process on the concatenated batches: per-item-mean losses make its gradient the mean of the two ranks' gradients
    cfg = O.Cfg(**CFG)
    m = build_model(cfg, O.formula_params(cfg, seed=100, with_tags=True)).train()
    opt = HidvaeAdamW(_groups(m, False), cosine=(1000, 7e-8)).prepare()

    def both(it):
        a, b = _batch(cfg, B, 700 + 2 * it, False), _batch(cfg, B, 700 + 2 * it + 1, False)
        return types.SimpleNamespace(x=torch.cat([a.x, b.x]))

    st = GraphedTrainStep(m, opt, [both(0)], gumbel_t=0.2, warmup=2)
    for it in range(steps):
        st([both(it)])
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    lr_travel = 2.8e-4 * steps
    for k in r0:
        d = (sd[k] - r0[k]).abs()
        # (Adam turns 1e-6-relative gradient differences -- here: another summation order over the batch -- into O(lr) steps wherever
        #  |g| ~ eps; the bulk must agree tightly, nothing may be off by more than a fraction of its possible travel)
        assert float(d.max()) <= 0.5 * lr_travel, (k, float(d.max()))
        assert float(d.median()) <= 0.01 * lr_travel, (k, float(d.median()))


def test_the_levels_early_optimizer_update_is_the_same_update():
    """Without a gradient exchange a tag level's head parameters take their AdamW update on that level's stream right after its
    backward (HidvaeAdamW.step_early through GraphedTrainStep's hook), the rest at the end: every parameter and every logged row after
    6 graphed steps must equal the run that updates everything at the end, bit for bit."""
    from hidvae_amd.optim import HidvaeAdamW
    from hidvae_amd.step import GraphedTrainStep
    cfg = O.Cfg(**CFG)
    out = []
    for early in (True, False):
        m = build_model(cfg, O.formula_params(cfg, seed=100, with_tags=True)).train()
        torch.manual_seed(1234)
        opt = HidvaeAdamW(_groups(m, True), cosine=(1000, 7e-8)).prepare()
        st = GraphedTrainStep(m, opt, [_batch(cfg, 128, 500, True)], gumbel_t=0.2, warmup=2)
        assert st.level_done_hook is not None and st.loss_grad_hint == 1.0
        assert getattr(m, "loss_grad_hint", None) is None and getattr(m, "_level_done_hook", None) is None  # armed only inside the stepper's forward
        if not early:
            st.level_done_hook = None
        rows = [st([_batch(cfg, 128, 500 + it, True)]).clone() for it in range(6)]
        assert st.graphs is not None and len(st.graphs) == 1
        out.append((torch.stack(rows).cpu(), {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}))
    assert torch.equal(out[0][0], out[1][0])
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k
