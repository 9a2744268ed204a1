"""One optimizer step of the tokenizer as a replayable HIP graph.

train_hidvae.train() and bench.py run the same sequence every iteration -- copy the batch into fixed input buffers, forward,
backward, (all-reduce,) AdamW -- 19 launches in the untagged case and about 170 with the tag heads, most of them a few
microseconds long.  Issued one by one from Python the step is host-bound (several ms with the tag heads on MI355X); captured
once and replayed it is bound by the device (DESIGN.md section 4 carries the current figures).  Nothing in the step synchronises with the host (losses, the mixup weight, the AdamW step
counter and the learning-rate schedule live on the device), which is what makes the capture legal.

Data parallel.  Over RCCL (backend "nccl") the gradient all-reduce is captured INSIDE the step's graph -- RCCL collectives are
graph nodes like any kernel -- so a data-parallel step is still ONE graph launch per optimizer step:

    graph[zero_grad, forward, backward, all-reduce(flat gradients), AdamW]

(round 2 kept the collective between two graphs: a second graph launch and two host-side enqueues per step, 45 us of fixed cost on
a 232 us step).  With a first bucket (HRqVae.dp_cut: the backward is split in two at the inputs of the decoder's tail and of the
loss launch, and the flat gradient buffer leads with the parameters the first half completes) the exchange overlaps the rest of
the backward, inside the same graph:

    graph[zero_grad, forward, backward part 1, all-reduce(bucket 1) on RCCL's stream || backward part 2, all-reduce(bucket 2), AdamW]

(DDP gives the reference the same thing by bucketing gradients in reverse registration order, train_hidvae.py:630-632,709.)
Backends whose collectives run on the host (gloo: CPU tests, several ranks on one GPU) cannot be captured; there the collectives
stay between graphs as before: graph[fwd, bwd part 1] -> all-reduce || graph[bwd part 2] -> all-reduce -> graph[AdamW], or
graph[fwd, bwd] -> all-reduce -> graph[AdamW].  HIDVAE_DP_GRAPH_COLLECTIVES=0/1 overrides the choice."""
import contextlib
import os
import types

import torch


class GraphedTrainStep:
    """step = GraphedTrainStep(model, opt, example_batches, dp=None, gumbel_t=0.2)
    row = step(batches)   # batches: list of `ga` batch records (x [, tags_emb, tags_indices]) of the example's shapes
    `row` is a device tensor [6] = (total loss, mean recon, mean rqvae, tag align, tag pred, tag accuracy) that the NEXT call
    overwrites: clone it to keep it.  The first calls run eagerly (allocator / kernel-attribute warm-up, and the randomness
    provider learns the step's dropout requests); the capture happens on call number `warmup + 1`."""

    def __init__(self, model, opt, example_batches, dp=None, gumbel_t=0.2, warmup=3, enabled=True, overlap=None):
        self.model, self.opt, self.dp, self.t = model, opt, dp, gumbel_t
        self.ga = len(example_batches)
        self.tagged = getattr(example_batches[0], "tags_emb", None) is not None
        self.static = []
        for b in example_batches:
            s = types.SimpleNamespace(x=torch.empty_like(b.x))
            if self.tagged:
                s.tags_emb, s.tags_indices = torch.empty_like(b.tags_emb), torch.empty_like(b.tags_indices)
            self.static.append(s)
        self.one = torch.ones((), device=example_batches[0].x.device)
        self.row = None
        self.calls, self.warmup, self.enabled = 0, warmup, enabled
        # overlap: None = by size.  Measured with one rank (bench.py --dist 1): the split costs ~50 us of fixed latency per step (a third
        # graph launch, a second collective, two more cross-stream edges): 0.330 vs 0.277 ms on the 4.6 MB core model, where a whole
        # 8-GPU all-reduce is of that order itself; on the 29 MB tagged model (1.94 vs 2.02 ms) it pays.  Hence: from 16 MB on.
        # PROVISIONAL: the threshold comes from one-rank runs where the collective moves no data; no N > 1 hardware measurement exists yet
        self.overlap = overlap
        self.graphs = None
        self.in_graph = False  # True once the collectives were captured inside the (single) step graph
        # the gradient every micro-batch's loss receives below (total.backward(gradient=1) through loss / ga): announced to the model
        # FOR THE DURATION OF THIS STEPPER'S OWN FORWARD CALLS (_armed) so the tag heads can run their backward right after their forward
        # (HRqVae.loss_grad_hint; HIDVAE_EARLY_HEADS=0 turns it off).  Outside those calls the model carries neither the hint nor the
        # hook: a user's own train-mode forward (another scaling of the loss, another accumulation count, a second stepper) has no side
        # effects on .grad or on the optimizer.
        self.loss_grad_hint, self.level_done_hook = None, None
        if self.tagged and hasattr(model, "n_layers") and os.environ.get("HIDVAE_EARLY_HEADS", "1") != "0":
            self.loss_grad_hint = float(torch.tensor(1.0, dtype=torch.float32) / self.ga) if self.ga > 1 else 1.0
            # ... and, with one micro-batch per step and no gradient exchange, a level's head parameters take their AdamW update on that
            # level's stream as soon as its backward is done (90 % of the tagged model's bytes leave the serial tail of the step)
            if (dp is None and self.ga == 1 and hasattr(opt, "step_early") and hasattr(model, "tag_predictors")
                    and os.environ.get("HIDVAE_EARLY_ADAMW", "1") != "0"):
                ranges = [opt.tensor_ranges_of(list(model.tag_predictors[i].parameters()) + list(model.tag_projectors[i].parameters()))
                          for i in range(model.n_layers)]
                if all(r is not None for r in ranges):
                    self.level_done_hook = lambda i: opt.step_early(ranges[i])

    @contextlib.contextmanager
    def _armed(self):
        """the model knows the loss gradient / the per-level optimizer hook only inside this block"""
        m = self.model
        saved = (getattr(m, "loss_grad_hint", None), getattr(m, "_level_done_hook", None))
        m.loss_grad_hint, m._level_done_hook = self.loss_grad_hint, self.level_done_hook
        try:
            yield
        finally:
            m.loss_grad_hint, m._level_done_hook = saved

    # -- the step, as plain code (this is what gets captured)
    def _overlapped(self):
        if not (self.dp is not None and self.ga == 1 and self.opt.flat_grads and getattr(self.opt, "n_first", 0) > 0
                and hasattr(self.model, "backward_rest")):
            return False
        if self.overlap is not None:
            return bool(self.overlap)
        env = os.environ.get("HIDVAE_DP_OVERLAP")
        if env is not None:
            return env == "1"
        return self.opt.grad_buffer.flat.numel() * 4 >= 16 * 1024 * 1024

    def _fwd_bwd(self):
        self.opt.zero_grad()
        total = None
        with self._armed():
            for s in self.static:
                out = self.model(s, gumbel_t=self.t)
                part = out.loss / self.ga if self.ga > 1 else out.loss
                total = part if total is None else total + part
        total.backward(gradient=self.one)
        if self.opt.flat_grads:
            self.opt.grad_buffer.seal()
        summary = self.model.last_summary
        self.row = summary if self.ga == 1 else torch.cat([total.detach().reshape(1), summary[1:]])

    def _part1(self):  # forward + the half of the backward that completes the first bucket (the decoder's tail)
        self.opt.zero_grad()
        self.model.dp_cut = True
        try:
            with self._armed():
                out = self.model(self.static[0], gumbel_t=self.t)
        finally:
            self.model.dp_cut = False
        out.loss.backward(gradient=self.one)
        self.opt.grad_buffer.seal(upto=self.opt.n_first)
        self.row = self.model.last_summary

    def _part2(self):  # the rest of the backward
        self.model.backward_rest()
        self.opt.grad_buffer.check_first_bucket_untouched()
        self.opt.grad_buffer.seal()

    def _exchange_overlapped(self, run1, run2):
        buf = self.opt.grad_buffer
        n1 = buf.numel_of_first(self.opt.n_first)
        run1()
        w1 = self.dp.allreduce_part(0, n1)            # on the wire while part 2 runs
        run2()
        w2 = self.dp.allreduce_part(n1, buf.flat.numel())
        for w in (w1, w2):
            if w is not None:
                w.wait()                               # (stream-side wait: the host does not block)
        return 1.0 / self.dp.world

    def _collectives_capturable(self):
        """RCCL collectives are stream operations and can be captured; host-side backends (gloo) cannot"""
        if self.dp is None:
            return False
        env = os.environ.get("HIDVAE_DP_GRAPH_COLLECTIVES")
        if env is not None:
            return env == "1"
        return self.dp.capturable()

    def _eager(self):
        if self._overlapped():
            self.opt.grad_scale = self._exchange_overlapped(self._part1, self._part2)
            self.opt.step()
            return
        self._fwd_bwd()
        if self.dp is not None:
            self.opt.grad_scale, _ = self.dp.allreduce()
        from . import _C
        _C.phase_mark("backward done")
        self.opt.step()
        if self.tagged and self.level_done_hook is not None:
            from .tagpath import join_tag_streams
            join_tag_streams(self.static[0].x.device)  # the levels' own optimizer updates ran on their streams, beside the core's backward
        _C.phase_mark("adamw done")

    def _capture_mode(self):
        """hipStreamCaptureMode of this step's captures.

        torch's default ("global") makes a potentially unsafe HIP call from ANY thread an error while this thread captures.  A process
        group's watchdog thread is such a caller: every 100 ms it polls the completion events (hipEventQuery) of collectives that were
        issued eagerly and that it has not yet seen complete -- the warm-up steps' collectives, for up to a polling period after the
        device went idle.  Under a global-mode capture that query returns hipErrorStreamCaptureUnsupported, the watchdog turns it into
        an exception nobody catches on its (C++-only) thread and the process aborts: round 3's intermittent "Fatal Python error: Aborted"
        of the first captured data-parallel step, signal raised on a thread without Python state (gpurun_out/r3f.log, r4a.log; the C++
        message went into pytest's captured stderr and was lost with the process).  tests/test_dp_gpu.py
        test_event_query_from_another_thread_* shows the mechanism in isolation, deterministically: the same query from a second thread
        fails a global-mode capture every time and is legal under a thread-local one.
        "thread_local" restricts the prohibition to the capturing thread -- which still must not (and does not) make unsafe calls --
        so other threads' event queries are none of this capture's business.  Kernel launches from autograd's device thread are
        captured as before: capture is a property of the stream, not of the thread."""
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                return "thread_local"
        except Exception:  # noqa: BLE001  (a build without distributed support has no watchdog either)
            pass
        return "global"

    def _graph(self, g, **kw):
        return torch.cuda.graph(g, capture_error_mode=self._capture_mode(), **kw)

    def _rejoin_forked_streams(self):
        """the current (capturing) stream waits for every stream this package may have forked work to"""
        from .ops import join_side
        from .tagpath import join_tag_streams
        join_side()
        join_tag_streams(self.static[0].x.device)
        comm = getattr(self.dp, "_comm", None) if self.dp is not None else None
        if comm:  # the own communicator's stream carries the asynchronous bucket exchanges
            torch.cuda.current_stream().wait_stream(comm.stream)

    def _reset_after_failed_capture(self):
        """a capture that raised part-way leaves host-side bookkeeping of a step that never ran: undo it before anything steps again"""
        from . import _C
        dev = self.static[0].x.device
        torch.cuda.synchronize()
        _C.take_pending_adamw(dev, owner=self.opt)       # a deferred schedule launch registered inside the failed capture
        self.opt._prepared = False                        # ... or issued inside it: the scalars were never computed
        self.opt._early = []
        self.opt.zero_grad()                              # drops .grad, un-seals the flat buffer, forgets first-bucket marks
        if getattr(self.model, "dp_cut", False):
            self.model.dp_cut = False
        if self.tagged:
            from .tagpath import join_tag_streams
            join_tag_streams(dev)

    def _capture(self):
        if self.dp is not None and self._collectives_capturable() and not getattr(self, "_in_graph_failed", False):
            g1 = torch.cuda.CUDAGraph()
            err = None
            try:
                with self._graph(g1):
                    try:
                        self._eager()  # the collectives become nodes of the graph (they fork to RCCL's stream and join back)
                    except RuntimeError as e:
                        # The capture itself must still END cleanly: a capture_end that fails ("capturing stream has unjoined work" --
                        # the level streams were forked into it) leaves torch's allocator and generator bookkeeping of this graph
                        # behind and the next capture takes the process down (seen on the GPU: "The graph should be registered to the
                        # state", thrown from ~CUDAGraph).  So: join every forked stream back, end the capture, discard the graph.
                        err = e
                        self._rejoin_forked_streams()
            except RuntimeError as e2:
                raise RuntimeError(
                    "capturing the data-parallel step with its collectives inside the graph failed and the capture could not be ended "
                    f"cleanly ({str(e2).splitlines()[0]}); this process cannot capture again -- restart with HIDVAE_DP_GRAPH_COLLECTIVES=0 "
                    "to keep the collectives between graphs") from (err or e2)
            if err is not None:
                # a communicator that refuses to be captured: keep the collectives between graphs instead of dying.  (This catches an
                # error, not a hang: bench.py's launcher / its per-rank time limit bound the run's wall time for that.)
                import sys
                print(f"[hidvae] capturing the collectives inside the step graph failed ({str(err).splitlines()[0]}); "
                      "keeping them between graphs", file=sys.stderr)
                del g1
                self._in_graph_failed = True
                self._reset_after_failed_capture()
                return self._capture()
            self.graphs = (g1,)
            self.in_graph = True
        elif self._overlapped():
            g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with self._graph(g1):
                self._part1()
                if self.tagged:  # the heads' early backward runs on the level streams: this graph ends here, so they join here
                    from .tagpath import join_tag_streams
                    join_tag_streams(self.static[0].x.device)
            with self._graph(g2, pool=g1.pool()):
                self._part2()
            self.opt.grad_scale = 1.0 / self.dp.world
            with self._graph(g3, pool=g1.pool()):
                self.opt.step()
            self.graphs = (g1, g2, g3)
        elif self.dp is not None:
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with self._graph(g1):
                self._fwd_bwd()
            self.opt.grad_scale = 1.0 / self.dp.world
            with self._graph(g2, pool=g1.pool()):
                self.opt.step()
            self.graphs = (g1, g2)
        else:
            g1 = torch.cuda.CUDAGraph()
            with self._graph(g1):
                self._eager()
            self.graphs = (g1,)

    def load(self, batches):
        if len(batches) != self.ga:
            raise RuntimeError(f"GraphedTrainStep was built for {self.ga} micro-batches, got {len(batches)}")
        for s, b in zip(self.static, batches):
            if b.x.shape != s.x.shape:
                raise RuntimeError(f"batch shape {tuple(b.x.shape)} differs from the captured {tuple(s.x.shape)}")
            # (a batch that was produced INTO the step's own input buffers -- input_buffers(), e.g. RandomBatches.next(out=...) -- is
            #  already where the graph reads it: no copy)
            if b.x.data_ptr() != s.x.data_ptr():
                s.x.copy_(b.x)
            if self.tagged:
                if b.tags_emb.data_ptr() != s.tags_emb.data_ptr():
                    s.tags_emb.copy_(b.tags_emb)
                if b.tags_indices.data_ptr() != s.tags_indices.data_ptr():
                    s.tags_indices.copy_(b.tags_indices)

    def input_buffers(self, micro=0):
        """The step's own input tensors of micro-batch `micro` (x [B, 768]; tags_emb, tags_indices when tagged): a loader that writes the
        next batch straight into them (a gather with `out=`, a host-to-device copy) and then passes them to __call__ saves the
        per-step copies (3 launches, 12.6 MB at the tagged B = 1024)."""
        return self.static[micro]

    def __call__(self, batches):
        self.load(batches)
        self.calls += 1
        if not self.enabled or self.calls <= self.warmup:
            self._eager()
            return self.row
        if self.graphs is None:
            torch.cuda.synchronize()
            self._capture()  # (a capture only records: replay it for this call's batch)
        if len(self.graphs) == 3:
            self._exchange_overlapped(self.graphs[0].replay, self.graphs[1].replay)
            self.graphs[2].replay()
            return self.row
        self.graphs[0].replay()
        if len(self.graphs) == 2:
            self.dp.allreduce()
            self.graphs[1].replay()
        return self.row
