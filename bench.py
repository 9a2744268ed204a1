#!/usr/bin/env python3
"""bench.py -- items/s of one HiD-VAE tokenizer TRAIN step (forward + backward + grad all-reduce + AdamW) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: the script starts N rank processes itself -- or is started by torch.distributed.run --, one rank per GPU over RCCL;
     per-rank batch fixed => weak scaling)

Workload at N=1 = BASELINE.json configs[1]: Amazon-Beauty-shaped synthetic items, 768-d unit-norm inputs,
hidden [512,256,128], D=32, 3 x 256 codebooks, ROTATION_TRICK, batch 1024 per GPU, amazon-gin hyper-parameters, WITH the tag heads
(every h-config of the reference feeds tags: train_hidvae.py:698-709); the untagged core step is carried beside it.
Inputs are resident in HBM before the timed region (a pool of pre-generated batches, one D2D copy per step).
One JSON line on rank 0 carrying `roofline` (dominant kernel, timed live with HIP events) and `cpu_baseline`
(the oracle's torch-CPU restatement of the same step, timed on this host for a bounded sample).
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
if "--cpu-point" not in sys.argv:  # (the CPU-baseline child never touches the GPU library)
    import hidvae_amd  # noqa: E402,F401  (before the first HIP call of the process: it sets the graph-queue default the runtime reads at init)

AMAZON = dict(commitment_weight=0.4, tag_alignment_weight=0.15, tag_prediction_weight=0.55, tag_class_counts=[38, 168, 348],
              use_focal_loss=True, focal_loss_params={"gamma_0": 2.7, "alpha_0": 0.24, "gamma_1": 2.7, "alpha_1": 0.24,
                                                      "gamma_2": 2.7, "alpha_2": 0.24},
              dropout_rate=0.4, alignment_temperature=0.1, sem_id_uniqueness_weight=1.5, sem_id_uniqueness_margin=0.0)
# reference configs/h_rqvae_kuairand.gin:20,30-32,39-40,50 (BASELINE config 3)
KUAIRAND = dict(commitment_weight=0.5, tag_alignment_weight=0.5, tag_prediction_weight=0.7, tag_class_counts=[37, 168, 353],
                use_focal_loss=True, focal_loss_params={"gamma_0": 2.5, "alpha_0": 0.25}, dropout_rate=0.25, alignment_temperature=0.08,
                sem_id_uniqueness_weight=0.5, sem_id_uniqueness_margin=0.5)
HPARAMS = {"amazon": AMAZON, "kuairand": KUAIRAND}
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # fp32-input MFMA = vector rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024, help="items per GPU per step")
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--codes", type=int, default=256)
    ap.add_argument("--tagged", type=int, default=1, help="1 (default): the HiD-VAE step with its tag heads -- every shipped h-config feeds "
                    "tags (reference train_hidvae.py:698-709); 0: the untagged core step")
    ap.add_argument("--hparams", choices=sorted(HPARAMS), default="amazon", help="hyper-parameters of configs/h_rqvae_<name>.gin")
    ap.add_argument("--also-configs", type=int, default=1, help="also time BASELINE configs 3 (kuairand-shaped tagged step at batch 2048) and 5 "
                    "(4 x 1024 codebooks at batch 4096, with the GPU k-means start-up), reported as `config3_step`, `config5_step`, `kmeans_init`")
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a HIP graph")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--pool", type=int, default=8, help="resident synthetic batches cycled through")
    ap.add_argument("--dist", type=int, default=0, help="take the multi-rank code path (process group, flat gradient buffer, "
                    "all-reduce between the two graphs) even with one rank: a rehearsal of the N>1 path on a one-GPU box")
    ap.add_argument("--kernels", type=int, default=1, help="time the dominant kernels for the roofline object (0 = skip, for profiling runs)")
    ap.add_argument("--also-large", type=int, default=8192, help="also time the same step at this per-GPU batch (reported as "
                    "`large_batch_step`; 0 = skip): shows the throughput-bound regime next to the latency-bound headline batch")
    ap.add_argument("--also-other", type=int, default=1, help="also time the other variant of the step (untagged core when --tagged 1, "
                    "tagged when --tagged 0), reported as `untagged_core_step` / `tagged_step`")
    ap.add_argument("--also-tagged", dest="also_other", type=int, help=argparse.SUPPRESS)  # round 1-2 name of --also-other (scratch/r3/*.sh)
    ap.add_argument("--time-limit", type=float, default=float(os.environ.get("HIDVAE_BENCH_TIME_LIMIT", "900")),
                    help="launcher only (--gpus N without torch.distributed.run): seconds after which the rank processes are terminated "
                         "and the launcher exits non-zero -- a rank stuck in a collective must not hold the job (0 = no limit)")
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each; the MEDIAN window is reported")
    return ap.parse_args()


def _hp(args):
    return HPARAMS[getattr(args, "hparams", "amazon")]


def build_model(args, device):
    import hidvae_amd  # noqa: F401
    from hidvae_amd.modules.h_rqvae import HRqVae
    from hidvae_amd.modules.quantize import QuantizeForwardMode
    torch.manual_seed(0)
    m = HRqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=args.codes, codebook_kmeans_init=False,
               codebook_normalize=True, codebook_mode=QuantizeForwardMode.ROTATION_TRICK, n_layers=args.levels, n_cat_features=0,
               tag_embed_dim=768, **{**_hp(args), "tag_class_counts": (_hp(args)["tag_class_counts"] + [500] * 8)[: args.levels]})
    # codebooks as k-means would leave them: residual-sized, spread (k-means itself is start-up work, not part of a step)
    with torch.no_grad():
        for i, layer in enumerate(m.layers):
            layer.embedding.weight.copy_((torch.rand_like(layer.embedding.weight) * 2 - 1) * (1.0 if i == 0 else 0.35 * 0.5 ** i))
    return m.to(device).train()


def param_groups(m, lr=2.8e-4, wd=0.015, pwd=0.015, tagged=False):
    """train_hidvae.py:537-561 of the reference"""
    groups = [{"params": list(m.encoder.parameters()) + list(m.decoder.parameters()), "lr": lr, "weight_decay": wd},
              {"params": [p for layer in m.layers for p in layer.parameters()], "lr": lr, "weight_decay": wd}]
    if tagged:
        for i in range(m.n_layers):
            plr, pw = lr * (1 + 0.1 * i), pwd / (1 + 0.2 * i)
            groups.append({"params": list(m.tag_predictors[i].parameters()), "lr": plr, "weight_decay": pw})
            groups.append({"params": list(m.tag_projectors[i].parameters()), "lr": plr, "weight_decay": pw})
    return groups


def synth_pool(args, device, rank):
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x = torch.nn.functional.normalize(torch.randn(args.pool, args.batch, 768, generator=g), dim=-1).to(device)
    te = ti = None
    if args.tagged:
        te = torch.randn(args.pool, args.batch, args.levels, 768, generator=g).to(device)
        classes = (_hp(args)["tag_class_counts"] + [500] * 8)[: args.levels]
        cols = [torch.randint(0, c, (args.pool, args.batch), generator=g) for c in classes]
        ti = torch.stack(cols, dim=-1)
        ti[torch.rand(ti.shape, generator=g) < 0.05] = -1
        ti = ti.to(device)
    return x, te, ti


def time_kernel(fn, launches=20, reps=20, side=None):
    """Average duration (us) of ONE launch of `fn`: `launches` back-to-back launches are captured into a HIP graph on
    torch's current stream (so the host is out of the picture) and the replays are bracketed by HIP events recorded on
    that same stream.  side: a stream announced with _C.register_ws_lane -- the launches are issued THERE inside the graph (forked
    from and joined back into the capturing stream), which is how the tag heads' level streams launch them."""
    def body():
        if side is None:
            fn()
            return
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            fn()
        cur.wait_stream(side)
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        if side is None:
            for _ in range(launches):
                fn()
        else:
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(launches):
                    fn()
            cur.wait_stream(side)
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * launches)


def kernel_rooflines(args, m, device):
    """Live per-kernel timings on the bench shapes.
    * gemm_tile16_kernel / gemm_direct_kernel / gemm_ring_bwd_kernel (fp32 MFMA roofline, 157.3 TFLOP/s): the largest GEMMs of the step;
    * rq_forward_kernel (HBM roofline, 8 TB/s): ALGORITHMIC bytes per item exactly as SURVEY.md 8(d) counts them for the tagged
      variant: read z 128 B, write ids 8L, per-level emb_out 128L, loss 4 = 540 B at L=3 (+ the codebooks 4*L*K*32 once per
      launch).  The launch also writes emb_sum and z (128 B each, consumed by the decoder and the backward): `bytes_moved`
      carries that 796 B/item figure, `traffic` the PMC measurement.  In exact fp32 this kernel is bound by the fp32 MFMA rate,
      not by HBM (2*L*K*32 FLOP per item); both fractions are reported."""
    from hidvae_amd import _C
    B, L, K = args.batch, args.levels, args.codes
    out = []
    x = torch.randn(B, 768, device=device)
    w0 = m.encoder.mlp[0].weight.detach()
    o0 = torch.empty(B, 512, device=device)
    a0 = torch.empty(B, 512, device=device)
    t = time_kernel(lambda: _C.gemm(_C.GEMM_NT, x, w0, out=o0, epilogue=_C.EPI_SILU, aux=a0))
    fl = 2.0 * B * 768 * 512
    out.append(dict(entry="hidvae_gemm_f32", kernel="gemm_tile16_kernel encoder layer 0: [B,768]x[768,512]^T + SiLU (exact ORDER-G chain, LDS-shared 32x64 tiles)", bound="mfma",
                    achieved=fl / t * 1e-6, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=fl / t * 1e-6 / MFMA_F32_PEAK_TF, traffic=None,
                    us=t, flops=fl))
    g = torch.randn(B, 512, device=device)
    gw = torch.empty(512, 768, device=device)
    t = time_kernel(lambda: _C.linear_bwd(g, x, None, False, dW=gw, bias=True))  # (the step's own launch: the first layer needs no dX)
    out.append(dict(kernel="gemm_ring_bwd_kernel encoder layer 0 backward, weight gradient only: dW [512,768] = g^T x, db (LDS-DMA ring, evenly dealt k-steps)", bound="mfma",
                    achieved=fl / t * 1e-6, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=fl / t * 1e-6 / MFMA_F32_PEAK_TF, traffic=None,
                    us=t, flops=fl))
    wd = m.decoder.weights()[-1].detach()  # [768, 512]: the widest Linear backward of the step (decoder's last layer)
    gd, xd = torch.randn(B, 768, device=device), torch.randn(B, 512, device=device)
    pre = torch.randn(B, 512, device=device)
    t = time_kernel(lambda: _C.linear_bwd(gd, xd, wd, True, _C.EPI_DSILU, pre))
    fl2 = 4.0 * B * 768 * 512
    dec = dict(kernel="gemm_ring_bwd_kernel decoder layer 3 backward: dW [768,512] = g^T x and dX = (g W) * silu'(pre) in one launch (LDS-DMA ring, 64x64 tiles, evenly dealt k-steps)",
               bound="mfma", achieved=fl2 / t * 1e-6, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=fl2 / t * 1e-6 / MFMA_F32_PEAK_TF, traffic=None,
               us=t, flops=fl2)
    # the tag heads' widest Linear backward (level 2's residual blocks, 691 <-> 768): dW + dX (through the ReLU -> Dropout gate of the layer
    # below, read off its output) + db in one launch
    gt, xt, wt, yt = torch.randn(B, 691, device=device), torch.randn(B, 768, device=device), torch.randn(691, 768, device=device) * 0.03, torch.rand(B, 768, device=device)
    lane = torch.cuda.Stream(device=device)
    _C.register_ws_lane(lane)  # (as the level streams are: a lane with a workspace of its own)
    t = time_kernel(lambda: _C.linear_bwd(gt, xt, wt, True, _C.EPI_DRELU, yt, bias=True, dx_scale=1.6), side=lane)
    fl3 = 4.0 * B * 691 * 768
    head = dict(kernel="gemm_ring_bwd_kernel tag-head layer backward 691 x 768 at B = 1024: dW = g^T x, dX = (g W) gated by the layer below, db "
                       "(LDS-DMA ring, four-wave workgroups two per CU, evenly dealt k-steps)",
                bound="mfma", achieved=fl3 / t * 1e-6, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=fl3 / t * 1e-6 / MFMA_F32_PEAK_TF, traffic=None,
                us=t, flops=fl3)
    (head if args.tagged else dec)["entry"] = "hidvae_linear_bwd"  # the stand-alone figure quoted beside the in-step family of the headline
    out += [dec, head]
    # the same layer at config 3's batch
    B2 = 2048
    gt2, xt2, yt2 = torch.randn(B2, 691, device=device), torch.randn(B2, 768, device=device), torch.rand(B2, 768, device=device)
    t = time_kernel(lambda: _C.linear_bwd(gt2, xt2, wt, True, _C.EPI_DRELU, yt2, bias=True, dx_scale=1.6), side=lane)
    fl4 = 4.0 * B2 * 691 * 768
    out.append(dict(kernel="gemm_ring_bwd_kernel tag-head layer backward 691 x 768 at B = 2048 (LDS-DMA ring, four-wave workgroups, evenly dealt k-steps): "
                           "dW, gated dX, db in one launch", bound="mfma", achieved=fl4 / t * 1e-6, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s",
                    frac=fl4 / t * 1e-6 / MFMA_F32_PEAK_TF, traffic=None, us=t, flops=fl4))
    # the same layer in the throughput regime (LDS-tiled kernel; corpus tokenisation and large-batch training run here)
    bigm = 1 << 16
    xb, ob, ab = torch.randn(bigm, 768, device=device), torch.empty(bigm, 512, device=device), torch.empty(bigm, 512, device=device)
    t = time_kernel(lambda: _C.gemm(_C.GEMM_NT, xb, w0, out=ob, epilogue=_C.EPI_SILU, aux=ab), launches=4, reps=5)
    flb = 2.0 * bigm * 768 * 512
    out.append(dict(kernel="gemm_f32_kernel<2,2,NT> encoder layer 0 at 65,536 rows (LDS-tiled, throughput regime)", bound="mfma",
                    achieved=flb / t * 1e-6, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=flb / t * 1e-6 / MFMA_F32_PEAK_TF, traffic=None,
                    us=t, flops=flb))
    del xb, ob, ab
    y = torch.randn(B, 32, device=device)
    tables = [layer.embedding.weight.detach() for layer in m.layers]
    cb, cc = _C.codebook_prepare(tables, [i == 0 for i in range(L)])
    t = time_kernel(lambda: _C.rq_forward(y, cb, cc, True, 3, True, 0.4))
    alg = (128 + 8 * L + 128 * L + 4) * B + 4 * L * K * 32
    out.append(dict(kernel="rq_forward_kernel (fused L-level VQ, code-split variant)", bound="hbm", achieved=alg / t * 1e-3,
                    peak=HBM_PEAK_GBS, unit="GB/s", frac=alg / t * 1e-3 / HBM_PEAK_GBS, traffic=None, us=t, algorithmic_bytes=alg,
                    bytes_moved=(128 + 8 * L + 128 * L + 128 + 128 + 4) * B + 4 * L * K * 32,
                    mfma_f32_frac=2.0 * B * L * K * 32 / t * 1e-6 / MFMA_F32_PEAK_TF))
    # the same VQ kernel at a corpus-sized launch (where it is throughput-, not latency-bound)
    big = 1 << 20
    yb = torch.randn(big, 32, device=device)
    t = time_kernel(lambda: _C.rq_forward(yb, cb, cc, True, 3, True, 0.4), launches=2, reps=5)
    algb = (128 + 8 * L + 128 * L + 4) * big + 4 * L * K * 32
    out.append(dict(kernel="rq_forward at 1,048,576 items (corpus-sized launch: rq_forward_pf32_kernel, split-bf16 prefilter on 32x32x16 MFMA + exact confirmation)",
                    bound="hbm", achieved=algb / t * 1e-3,
                    peak=HBM_PEAK_GBS, unit="GB/s", frac=algb / t * 1e-3 / HBM_PEAK_GBS, traffic=None, us=t, algorithmic_bytes=algb,
                    bytes_moved=(128 + 8 * L + 128 * L + 128 + 128 + 4) * big + 4 * L * K * 32,
                    fp32_equivalent_mfma_frac=2.0 * big * L * K * 32 / t * 1e-6 / MFMA_F32_PEAK_TF, items_per_s=big / t * 1e6))
    # the ids-only form of the same search (HSemanticIdTokenizer.precompute_corpus_ids: only the ids leave the launch), priced on ITS
    # algorithmic bytes: read z 128 B + write ids 8 L per item
    t = time_kernel(lambda: _C.rq_ids(yb, cb, cc, True), launches=2, reps=5)
    algi = (128 + 8 * L) * big + 4 * L * K * 32
    out.append(dict(kernel="rq_forward ids-only at 1,048,576 items (rq_forward_pf32_kernel<.., IDS>: eval search, no output rows, no winner fetch after the last level)",
                    bound="hbm", achieved=algi / t * 1e-3, peak=HBM_PEAK_GBS, unit="GB/s", frac=algi / t * 1e-3 / HBM_PEAK_GBS, traffic=None, us=t,
                    algorithmic_bytes=algi, bytes_moved=algi, fp32_equivalent_mfma_frac=2.0 * big * L * K * 32 / t * 1e-6 / MFMA_F32_PEAK_TF,
                    items_per_s=big / t * 1e6))
    # the loader's batch formation (ResidentItemData.gather_into -> hidvae_gather_rows): rows of the resident item tables picked by the
    # batch's ids, all tables in one launch; algorithmic bytes = every gathered byte read once and written once + the ids
    n_items = 8 * B
    gx, gte = torch.randn(n_items, 768, device=device), torch.randn(n_items, L, 768, device=device)
    gti = torch.randint(0, 38, (n_items, L), device=device)
    gidx = torch.randperm(n_items, device=device)[:B].contiguous()
    gout = [torch.empty(B, 768, device=device), torch.empty(B, L, 768, device=device), torch.empty(B, L, dtype=torch.int64, device=device)]
    t = time_kernel(lambda: _C.gather_rows(gidx, [gx, gte, gti], gout))
    algg = B * (2 * (768 * 4 * (1 + L) + 8 * L) + 8)
    out.append(dict(entry="hidvae_gather_rows", kernel=f"gather_rows_kernel: a tagged batch of {B} items gathered from the resident tables (x, tags_emb, tags_indices) "
                                                       "into the step's input buffers, one launch",
                    bound="hbm", achieved=algg / t * 1e-3, peak=HBM_PEAK_GBS, unit="GB/s", frac=algg / t * 1e-3 / HBM_PEAK_GBS, traffic=None, us=t,
                    algorithmic_bytes=algg))
    del gx, gte, gti
    return out


def step_timeline(stepper, pool_batch, device, replays=9):
    """In-step duration of every launch of the replayed train step.  HIP events cannot be recorded inside a captured graph on
    ROCm, so the step is re-captured with a device-timestamp launch (hidvae_timestamp: wall_clock64, 10 ns ticks) before and after
    every C-ABI launch, on that launch's own stream; bracket minus the empty bracket = the launch's duration plus one dependent-
    launch gap, i.e. what it costs the step (the brackets of the untagged step sum to the un-stamped step time within 1 %).
    -> list of dict(entry, us, M, N, K, flops) in launch order."""
    from hidvae_amd import _C
    inner = stepper._fwd_bwd

    def stamped():
        for _ in range(5):
            _C.stamp_empty_bracket()
        inner()

    graphs = stepper.graphs
    stepper.graphs, stepper._fwd_bwd = None, stamped
    stamps = _C.stamps_begin(device)
    try:
        stepper([pool_batch(0)])  # captures the stamped step and replays it once
    finally:
        _C.stamps_end()
        stepper._fwd_bwd = inner
    # every launch's bracket is the MEDIAN over the replays: with three lanes sharing the chip a single replay's brackets swing by a
    # factor of two from one replay to the next (a bracket also holds whatever the lane waited for inside it)
    per_replay, empties = [], []
    for i in range(max(1, replays)):
        stepper([pool_batch(1 + i)])
        torch.cuda.synchronize()
        r, e = stamps.rows_with_dims()
        per_replay.append(r)
        empties.append(e)
    stepper.graphs = graphs  # back to the un-stamped graph
    mid = lambda v: sorted(v)[len(v) // 2]
    rows = [(per_replay[0][k][0], mid([r[k][1] for r in per_replay]), per_replay[0][k][2]) for k in range(len(per_replay[0]))]
    empty = mid(empties)
    out = []
    for name, us, dims in rows:
        r = dict(entry=name, us=us)
        if dims is not None:
            r.update(M=dims[0], N=dims[1], K=dims[2], flops=dims[3])
        out.append(r)
    return out, empty


def summarize_timeline(rows):
    """-> (per-entry-point table sorted by in-step time, the single dominant launch)"""
    agg = {}
    for r in rows:
        a = agg.setdefault(r["entry"], dict(entry=r["entry"], launches=0, us=0.0, flops=0.0))
        a["launches"] += 1
        a["us"] += r["us"]
        a["flops"] += r.get("flops", 0.0)
    total = sum(r["us"] for r in rows)
    table = sorted(agg.values(), key=lambda a: -a["us"])
    for a in table:
        a["share"] = a["us"] / total
        if a["flops"]:
            a["tflops"] = a["flops"] / a["us"] * 1e-6
            a["mfma_f32_frac"] = a["tflops"] / MFMA_F32_PEAK_TF
    top = dict(table[0])  # the entry point (= kernel family) with the largest share of the step, as a rocprof per-kernel stats row
    top["us_per_launch"] = top["us"] / top["launches"]
    shapes = sorted({(r.get("M"), r.get("N"), r.get("K")) for r in rows if r["entry"] == top["entry"] and "M" in r}, key=lambda t: -t[0] * t[1] * t[2])
    top["shapes"] = shapes
    return table, top, total


def describe_launch(r):
    what = {"hidvae_gemm_f32": "forward Linear layers: gemm_tile16 / gemm_directL16 / gemm_direct16 kernels, fp32 MFMA, one exact ORDER-G chain per output",
            "hidvae_linear_bwd": "one-launch Linear backward dW = g^T x + dX = g W (+ db): gemm_ring_bwd_kernel (LDS-DMA ring, layers from 0.2 GFLOP) / gemm_pair16 (narrow layers), fp32 MFMA",
            "hidvae_bottleneck_fwd": "fused middle launch: encoder[-2:] + L-level RQ + decoder[:2], fp32 MFMA"}.get(r["entry"], "")
    shapes = "; ".join(f"{m}x{n}x{k}" for m, n, k in r.get("shapes", [])[:8])
    return f"{r['entry']} x{r['launches']} launches per step ({what}{'; MxNxK: ' + shapes if shapes else ''})"


def pmc_traffic():
    """{kernel-name prefix: bytes per launch} from profiles/*_pmc_hbm_traffic.csv (FETCH_SIZE doubled as the guide prescribes for
    gfx950, + WRITE_SIZE; both in KB in the file).  The counters need their own rocprofv3 passes, so they cannot be taken live
    here; absent file -> {} and `traffic` stays null."""
    import csv
    import glob
    out = {}
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_hbm_traffic.csv")))
    if not files:
        return out
    try:
        for row in csv.DictReader(open(files[-1])):
            out[row["bench_kernel_prefix"]] = (2.0 * float(row["FETCH_SIZE_KB"]) + float(row["WRITE_SIZE_KB"])) * 1024.0
    except Exception:  # noqa: BLE001  a malformed evidence file must not break the bench
        return {}
    return out


def _cpu_steps(args, budget_s, threads, warm=2):
    from oracle import torch_oracle as O
    torch.set_num_threads(threads)
    cfg = O.Cfg(n_layers=args.levels, codebook_size=args.codes, codebook_mode=O.ROTATION,
                **{**_hp(args), "tag_class_counts": (_hp(args)["tag_class_counts"] + [500] * 8)[: args.levels]})
    P = O.formula_params(cfg, seed=100, with_tags=bool(args.tagged))
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    opt = torch.optim.AdamW(list(Pg.values()), lr=2.8e-4, weight_decay=0.015)
    x, te, ti = O.formula_batch(cfg, args.batch, seed=7, tagged=bool(args.tagged))
    bn = None
    if args.tagged:
        bn = {f"tag_projectors.{i}.1.running_{s}": (torch.zeros(512) if s == "mean" else torch.ones(512))
              for i in range(args.levels) for s in ("mean", "var")}
    rand = O.TorchRand(cfg.mixup_alpha)
    n, first = 0, None
    while True:
        opt.zero_grad()
        out = O.forward(Pg, cfg, x, te, ti, gumbel_t=0.2, training=True, rand=rand, bn_buffers=bn)
        out["loss"].backward()
        opt.step()
        n += 1
        if n == warm:
            first = time.perf_counter()  # warm-up steps
        if first is not None and n > warm and (time.perf_counter() - first > budget_s or n >= warm + 200):
            break
    return n - warm, time.perf_counter() - first


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _usable_cores():
    """Cores this process may actually run on: the smaller of the affinity mask and the cgroup CPU quota (a GPU box shows all of the
    host's logical CPUs in os.cpu_count() while granting a share of them; threads beyond the share only queue)."""
    n = os.cpu_count() or 8
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _cpu_point_child(argv):
    """`python bench.py --cpu-point THREADS BUDGET <workload flags>`: one thread-count point of the CPU baseline in a process of its own (no
    GPU use), so the parent can bound it -- all cores of a 100+-core host take ~20 s per step on these small-op graphs"""
    i = argv.index("--cpu-point")
    threads, budget = int(argv[i + 1]), float(argv[i + 2])
    sys.argv = [argv[0]] + argv[1:i] + argv[i + 3:]
    a = parse()
    steps, dt = _cpu_steps(a, budget, threads, warm=1)
    print(json.dumps({"threads": threads, "steps": steps, "seconds": dt}), flush=True)


def cpu_baseline(args, budget_s, all_cores=True):
    """The oracle's torch-CPU restatement of the SAME step (fwd + bwd + torch AdamW, reference op sequence incl. the
    O(B^2) p_unique_ids), timed on this host.  This is the checker being timed as a baseline, never the product.
    BASELINE.md section 3: CPU model and core count stated; points at 1 thread, 8 threads (the survey container's count: BASELINE.md
    section 2 holds the true reference code's numbers at 8) and all cores.  Small-op PyTorch does not scale with cores, so the BEST
    point is `value`; the all-core point runs in a child process under a time limit."""
    import subprocess
    host = os.cpu_count() or 8
    ncpu = _usable_cores()
    points, notes, best = [], [], None

    def add(th, steps, dt, note=""):
        nonlocal best
        ips = args.batch * steps / dt if steps > 0 and dt > 0 else 0.0
        points.append(dict(threads=th, items_per_s=ips, steps=steps, seconds=dt))
        notes.append(f"{th} threads: {ips:.0f} items/s ({steps} steps, {dt:.1f} s){note}")
        if ips > 0 and (best is None or ips > best[0]):
            best = (ips, th)

    steps, dt = _cpu_steps(args, budget_s * 0.6, min(8, ncpu))
    add(min(8, ncpu), steps, dt)
    steps, dt = _cpu_steps(args, budget_s * 0.3, 1, warm=1)
    add(1, steps, dt)
    if ncpu > 8 and all_cores:
        flags = ["--batch", str(args.batch), "--levels", str(args.levels), "--codes", str(args.codes), "--tagged", str(args.tagged),
                 "--hparams", getattr(args, "hparams", "amazon")]
        limit = max(15.0, 2.0 * budget_s)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-point", str(ncpu), str(budget_s * 0.3)] + flags,
                               capture_output=True, text=True, timeout=limit, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
            rec = json.loads(r.stdout.strip().splitlines()[-1])
            add(ncpu, rec["steps"], rec["seconds"])
        except subprocess.TimeoutExpired:
            points.append(dict(threads=ncpu, items_per_s=None, steps=0, seconds=limit))
            notes.append(f"{ncpu} threads (all usable cores of {host}): not finished within {limit:.0f} s (1 warm-up + 1 step)")
        except Exception as e:  # noqa: BLE001  the all-core point is a report, not a requirement
            notes.append(f"{ncpu} threads (all cores): failed ({type(e).__name__})")
    return dict(value=best[0], unit="items/s", cores=best[1], kind="port", cpu_model=_cpu_model(), host_cores=host, usable_cores=ncpu, points=points,
                sample=f"full train steps (fwd+bwd+AdamW) of the oracle's torch-CPU restatement at B={args.batch}, "
                       f"{'tagged' if args.tagged else 'untagged'}, {getattr(args, 'hparams', 'amazon')} hyper-parameters, warm-up then a bounded "
                       "sample per thread count: " + "; ".join(notes))


def kmeans_init_times(device, budget_s=4.0):
    """BASELINE config 5 / SURVEY 8(d): the codebook start-up -- k-means on 20,000 x 32 residual-shaped rows to K = 256 and K = 1024
    (reference init/kmeans.py:34-77, Lloyd to convergence) -- on the GPU kernels, and the CPU restatement (oracle/kmeans_oracle.py) for a
    bounded number of Lloyd iterations beside it."""
    import numpy as np
    from hidvae_amd.init.kmeans import Kmeans
    from oracle import kmeans_oracle as KO
    g = torch.Generator(device="cpu").manual_seed(4321)
    x_cpu = torch.randn(20000, 32, generator=g) * 0.35
    x = x_cpu.to(device)
    out = []
    for K in (256, 1024):
        init = np.random.RandomState(7).choice(20000, K, replace=False)
        Kmeans(k=K, max_iters=2, init_indices=init).run(x)  # warm-up (kernel load)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        km = Kmeans(k=K, init_indices=init)
        km.run(x)
        torch.cuda.synchronize()
        gpu_s = time.perf_counter() - t0
        torch.set_num_threads(min(8, os.cpu_count() or 8))
        t0, it = time.perf_counter(), 0
        c = x_cpu[torch.as_tensor(init)].clone()
        while it < 3 and (it == 0 or time.perf_counter() - t0 < budget_s / 2):
            c, _ = KO.lloyd_iteration(x_cpu, c, reseed=lambda cl: 0)
            it += 1
        cpu_it = (time.perf_counter() - t0) / it
        out.append(dict(N=20000, D=32, K=K, gpu_seconds=gpu_s, gpu_iterations=km.n_iter, gpu_seconds_per_iteration=gpu_s / max(1, km.n_iter),
                        cpu_seconds_per_iteration=cpu_it, cpu_iterations_timed=it, cpu_threads=min(8, os.cpu_count() or 8),
                        cpu_seconds_same_iterations=cpu_it * km.n_iter))
    return out


def run_workload(args, device, rank, world, dist):
    """Build the model + optimizer, capture one full train step in a HIP graph, time K replays.  -> (seconds, model, info)"""
    from hidvae_amd.optim import HidvaeAdamW
    m = build_model(args, device)
    multi = dist is not None  # the data-parallel path (also taken with one rank under --dist 1)
    opt = HidvaeAdamW(param_groups(m, tagged=bool(args.tagged)), cosine=(400000, 7e-8), flat_grads=multi,
                      first_bucket=m.dp_first_bucket(args.batch) if multi else None).prepare()
    dp = None
    if multi:
        from hidvae_amd.parallel import DataParallel
        dp = DataParallel(m, opt.grad_buffer)
        dp.always = bool(args.dist)
        dp.broadcast_parameters(0)  # replicas start identical (DDP semantics)
    pool_x, pool_te, pool_ti = synth_pool(args, device, rank)

    def pool_batch(i):  # one prepared batch of the pool (warm-up, capture and the stamped timeline copy it into the step's inputs; the timed steps gather theirs, see step())
        b = types.SimpleNamespace(x=pool_x[i % args.pool])
        if args.tagged:
            b.tags_emb, b.tags_indices = pool_te[i % args.pool], pool_ti[i % args.pool]
        return b

    # the product's own step object (hidvae_amd/step.py, what train_hidvae.train() runs): 3 eager calls, then capture + replay;
    # under DP: graph[fwd+bwd] -> ONE RCCL all-reduce of the flat gradient buffer (1/world folded into AdamW) -> graph[AdamW]
    from hidvae_amd.step import GraphedTrainStep
    stepper = GraphedTrainStep(m, opt, [pool_batch(0)], dp=dp, gumbel_t=0.2, warmup=3, enabled=bool(args.graph))
    use_graph = bool(args.graph)
    try:
        for i in range(4):  # 3 eager warm-up calls + the capturing call, outside the timed region
            stepper([pool_batch(i)])
    except Exception as e:  # noqa: BLE001  report and run eagerly rather than die
        import traceback
        print(f"[bench] graph capture failed ({type(e).__name__}); running eagerly\n" + "".join(traceback.format_exc().splitlines(True)[-14:]),
              file=sys.stderr)
        stepper.enabled, stepper.graphs, use_graph = False, None, False
        torch.cuda.synchronize()

    # The timed step forms its batch as train_hidvae.train() does: the sampler's ids pick rows of the resident item tables (here: the
    # synthetic pool as one table of pool * batch items, a fixed shuffle of it cut into batches) and ResidentItemData.gather_into
    # writes them straight into the step's input buffers -- one hidvae_gather_rows launch, on the step's stream, inside the timed region
    from hidvae_amd.data.items import ResidentItemData
    table = ResidentItemData(pool_x.reshape(-1, pool_x.shape[-1]), pool_te.reshape(-1, *pool_te.shape[2:]) if args.tagged else None,
                             pool_ti.reshape(-1, pool_ti.shape[-1]) if args.tagged else None)
    shuffle = torch.randperm(len(table), generator=torch.Generator().manual_seed(99 + rank)).to(device)
    ids = [shuffle[j * args.batch:(j + 1) * args.batch].contiguous() for j in range(args.pool)]

    def step(i):
        stepper([table.gather_into(ids[i % args.pool], stepper.input_buffers())])

    for i in range(args.warmup):
        step(i)
    # `windows` timed regions of EXACTLY `steps` steps each, every one bracketed by barrier + synchronize on both sides and reduced
    # with MAX over ranks; the median window is the reported one (a single 5 ms sample is at the mercy of one clock ramp)
    windows, n = [], args.warmup
    for _ in range(max(1, getattr(args, "windows", 1))):
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(n + i)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        dt = time.perf_counter() - t0
        if multi:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        windows.append(dt)
        n += args.steps
    dt = sorted(windows)[len(windows) // 2]
    final_loss = float(stepper.row[0]) if stepper.row is not None else float("nan")
    return dt, m, dict(hip_graph=bool(use_graph), final_loss=final_loss, windows_ms_per_step=[w / args.steps * 1e3 for w in windows],
                       stepper=stepper, pool_batch=pool_batch)



def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, child=None, env=None, time_limit=None):
    """`python bench.py --gpus N` with no launcher around it: start N rank processes, one per device, and relay rank 0's JSON line.

    Runs BEFORE anything in this process touches the GPU (the parent never does: it only waits).  Every child is this script again
    with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set -- exactly the environment `python -m torch.distributed.run
    --nproc-per-node N` gives it, so both ways of starting the N > 1 bench run the same code.  (The reference gets its ranks from
    `accelerate launch`, train_hidvae.py:186-189.)  Rank 0's stdout is relayed to ours, the other ranks' stdout goes to stderr; the
    exit code is the first non-zero child code; when one rank dies the others are terminated (by PID) instead of left waiting in a
    collective.  time_limit (seconds): a run that is still going then -- a rank stuck in a collective or in a capture, which no
    exception handler inside the rank can catch -- is terminated (SIGTERM, then SIGKILL, by PID: the children are fresh processes of
    this launcher, nothing is re-executed) and the launcher returns 124, the code `timeout` uses.  `child` replaces the command (tests)."""
    import subprocess
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("MASTER_PORT", str(_free_port()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = list(child) if child is not None else [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), HIDVAE_BENCH_CHILD="1")
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading

    def relay():  # rank 0 prints the one JSON line (and nothing else) on stdout
        for line in procs[0].stdout:
            sys.stdout.write(line.decode() if isinstance(line, bytes) else line)
            sys.stdout.flush()

    pump = threading.Thread(target=relay, daemon=True)
    pump.start()
    rc = 0
    deadline = time.monotonic() + time_limit if time_limit else None
    try:
        pending = list(procs)
        while pending:
            if deadline is not None and time.monotonic() > deadline:
                print(f"[bench] the {n}-rank run exceeded its time limit of {time_limit:.0f} s "
                      f"(ranks still running: {[procs.index(p) for p in pending]}); terminating them", file=sys.stderr, flush=True)
                for q in pending:
                    q.terminate()
                t_kill = time.monotonic() + 10
                while any(q.poll() is None for q in pending) and time.monotonic() < t_kill:
                    time.sleep(0.05)
                rc = 124
                break
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:  # a dead rank leaves the others in a collective that never completes
                        q.terminate()
            if pending:
                time.sleep(0.05)
        pump.join(timeout=10)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    if "--cpu-point" in sys.argv:
        return _cpu_point_child(sys.argv)
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one.  (device_count() may call hipGetDeviceCount here; that is harmless because this process only spawns
        # children and never execs or launches anything itself)
        if os.environ.get("HIDVAE_DIST_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < args.gpus:
            print(f"[bench] --gpus {args.gpus} but only {torch.cuda.device_count()} device(s) visible (RCCL wants one device per rank)", file=sys.stderr)
            sys.exit(2)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], time_limit=args.time_limit or None))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and args.time_limit:
        # a rank of a multi-rank run (our launcher's child, or torch.distributed.run's): a hang in a collective or in a graph capture
        # raises nothing, so nothing below could handle it -- after the limit this prints every thread's stack and ends the process
        # with a non-zero code (the launcher / the elastic agent then stops the other ranks)
        import faulthandler
        faulthandler.dump_traceback_later(args.time_limit, exit=True)
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        if "WORLD_SIZE" in os.environ and args.gpus > 1:
            print(f"[bench] --gpus {args.gpus} disagrees with WORLD_SIZE={world} of the launcher", file=sys.stderr)
            sys.exit(2)
    if os.environ.get("HIDVAE_DIST_BACKEND", "nccl") != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1 or args.dist:
        import torch.distributed as dist
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        # "nccl" is RCCL on ROCm.  HIDVAE_DIST_BACKEND=gloo rehearses the N > 1 path with several ranks on ONE GPU (RCCL refuses two
        # ranks on one device); then every rank uses device LOCAL_RANK % device_count
        backend = os.environ.get("HIDVAE_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
        if dist.get_world_size() != world:
            raise RuntimeError(f"process group reports {dist.get_world_size()} ranks, the environment said {world}")
        world = dist.get_world_size()  # n_gpus of the line below = what the communicator (RCCL) reports, not what a flag asked for

    dt, m, info = run_workload(args, device, rank, world, dist)
    use_graph, final_loss = info["hip_graph"], info["final_loss"]
    single = world == 1 and dist is None
    timeline = None
    if rank == 0 and single and use_graph and args.kernels:
        timeline = step_timeline(info["stepper"], info["pool_batch"], device)

    def variant(vargs, label):
        """the step's other variant (or the large-batch one) as a sub-object of the line: time, launches, in-step shares, own roofline"""
        vdt, _, vinfo = run_workload(vargs, device, rank, world, dist)
        ex = dict(value=vargs.batch * vargs.steps / vdt, unit="items/s", ms_per_step=vdt / vargs.steps * 1e3, steps=vargs.steps,
                  batch=vargs.batch, hip_graph=vinfo["hip_graph"], windows_ms_per_step=vinfo["windows_ms_per_step"], workload=label)
        if vinfo["hip_graph"] and args.kernels and dist is None and vargs.batch <= 4096:
            vrows, _ = step_timeline(vinfo["stepper"], vinfo["pool_batch"], device)
            vtable, vtop, _ = summarize_timeline(vrows)
            ex["launches"] = len(vrows)
            ex["in_step_by_entry_point"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in a.items()} for a in vtable[:8]]
            ex["roofline"] = dict(bound="mfma", kernel=describe_launch(vtop), us_per_launch=vtop["us_per_launch"], peak=MFMA_F32_PEAK_TF,
                                  unit="TFLOP/s", achieved=vtop.get("flops", 0.0) / vtop["us"] * 1e-6,
                                  frac=vtop.get("flops", 0.0) / vtop["us"] * 1e-6 / MFMA_F32_PEAK_TF, traffic=None, share_of_step=vtop["share"])
        return ex, vargs

    other_extra = other_args = None
    if single and args.also_other:
        oa = argparse.Namespace(**{**vars(args), "tagged": 0 if args.tagged else 1, "steps": max(20, args.steps // (1 if args.tagged else 4)),
                                   "warmup": max(5, args.warmup // 2)})
        other_extra, other_args = variant(oa, "untagged core step: encoder, L-level RQ, decoder, reconstruction + commitment losses (no tag heads)" if args.tagged
                                          else "same shapes + tag heads (projector, InfoNCE, predictor, focal+mixup), amazon gin hyper-parameters")
        if args.cpu_seconds > 0:
            other_extra["cpu_baseline"] = cpu_baseline(other_args, max(4.0, args.cpu_seconds / 2), all_cores=False)
    large_extra = None
    if single and args.also_large and args.kernels:
        la = argparse.Namespace(**{**vars(args), "tagged": 0, "batch": args.also_large, "steps": max(20, args.steps // 5), "warmup": 5, "pool": 2})
        large_extra, _ = variant(la, f"untagged core step at batch {args.also_large} (the per-GPU shard of BASELINE config 4): the throughput-bound regime")

    config_extras = {}
    if single and args.also_configs and args.kernels:
        c3 = argparse.Namespace(**{**vars(args), "tagged": 1, "batch": 2048, "hparams": "kuairand", "levels": 3, "codes": 256,
                                   "steps": max(20, args.steps // 4), "warmup": 5, "pool": 4})
        config_extras["config3_step"], _ = variant(c3, "BASELINE config 3: KuaiRand-shaped tagged step (configs/h_rqvae_kuairand.gin hyper-parameters), "
                                                       "768-d + tag embeddings, 3x256, batch 2048")
        c5 = argparse.Namespace(**{**vars(args), "tagged": 0, "batch": 4096, "levels": 4, "codes": 1024, "steps": max(20, args.steps // 4),
                                   "warmup": 5, "pool": 2})
        config_extras["config5_step"], _ = variant(c5, "BASELINE config 5: 4 levels x 1024 codes, batch 4096 per GPU, untagged core step (streamed RQ: the "
                                                       "codebooks do not fit LDS)")
        config_extras["kmeans_init"] = kmeans_init_times(device)

    if rank == 0:
        if not args.kernels:  # profiling run of the step only (rocprofv3 timelines): no roofline object
            print(json.dumps({"value": args.batch * world * args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "note": "--kernels 0"}),
                  flush=True)
            if dist is not None:
                if world > 1:
                    dist.barrier()
                dist.destroy_process_group()
            return
        ks = kernel_rooflines(args, m, device)
        pmc = pmc_traffic()
        for k in ks:  # HBM-side bytes per launch from the committed rocprofv3 --pmc passes of exactly these launches (profiles/);
            for key, val in pmc.items():  # a kernel whose name no longer matches a CSV row keeps traffic = null
                if k["kernel"].startswith(key):
                    k["traffic"] = val
        # `roofline` = the kernel family with the largest IN-STEP time (device timestamps inside the replayed graph); the stand-alone
        # warm figure of the same shape (back-to-back launches on identical operands) is carried separately as us_warm / frac_warm
        roof, in_step = None, None
        if timeline is not None:
            rows, empty_us = timeline
            table, top, total = summarize_timeline(rows)
            gemm_rows = [r for r in rows if "flops" in r]
            in_step = dict(launches=len(rows), sum_us=total, empty_bracket_us=empty_us,
                           gemm_class=dict(launches=len(gemm_rows), us=sum(r["us"] for r in gemm_rows), flops=sum(r["flops"] for r in gemm_rows)),
                           by_entry_point=[{k: (round(v, 4) if isinstance(v, float) else v) for k, v in a.items()} for a in table[:24]],
                           note="brackets of launches on different streams overlap (the level branches run side by side): the sum exceeds the step time")
            # the MFMA roofline is priced on the GEMM-class family with the largest in-step time (a LayerNorm launch has no FLOP count)
            fam = next((a for a in table if a.get("flops")), None)
            top = dict(fam) if fam is not None else top
            top["us_per_launch"] = top["us"] / top["launches"]
            top["shapes"] = sorted({(r.get("M"), r.get("N"), r.get("K")) for r in rows if r["entry"] == top["entry"] and "M" in r}, key=lambda t: -t[0] * t[1] * t[2])
            ach = top.get("flops", 0.0) / top["us"] * 1e-6
            warm = next((k for k in ks if k.get("entry") == top["entry"]), None)  # the family's largest launch, stand-alone and warm
            roof = dict(kernel=describe_launch(top), bound="mfma", achieved=ach, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=ach / MFMA_F32_PEAK_TF,
                        traffic=None, us=top["us_per_launch"], share_of_step=top["us"] / total,
                        us_warm=warm["us"] if warm else None, frac_warm=warm["frac"] if warm else None)
            # the family's LARGEST launch (most FLOPs; the longest of them in-step): the one `warm` times stand-alone and the PMC pass
            # measured -- their figures are attached only if the shapes agree
            fam_rows = [r for r in rows if r["entry"] == top["entry"]]
            most = max(r.get("flops", 0.0) for r in fam_rows)
            big = max((r for r in fam_rows if r.get("flops", 0.0) >= 0.999 * most), key=lambda r: r["us"])
            same = warm is not None and abs(warm.get("flops", -1.0) - big.get("flops", 0.0)) <= 1e-6 * max(1.0, big.get("flops", 0.0))
            roof["traffic"] = warm.get("traffic") if same else None  # PMC bytes per launch of the family's largest launch (profiles/*_pmc_hbm_traffic.csv)
            roof["largest_launch"] = dict(MxNxK=[big.get("M"), big.get("N"), big.get("K")], us=big["us"],
                                          achieved=big.get("flops", 0.0) / big["us"] * 1e-6, frac=big.get("flops", 0.0) / big["us"] * 1e-6 / MFMA_F32_PEAK_TF,
                                          traffic=warm.get("traffic") if same else None, us_warm=warm.get("us") if same else None)
            # the whole step against the fp32 MFMA peak: every GEMM-class FLOP of the step over the step's wall time
            roof["whole_step"] = dict(flops=in_step["gemm_class"]["flops"], achieved=in_step["gemm_class"]["flops"] / (dt / args.steps) * 1e-12,
                                      frac=in_step["gemm_class"]["flops"] / (dt / args.steps) * 1e-12 / MFMA_F32_PEAK_TF)
        if roof is None:
            roof = dict(ks[0])
        line = {
            "metric": "item-embeddings/sec HiD-VAE train step, 768-d in, 3x256 codebooks",
            "value": args.batch * world * args.steps / dt, "unit": "items/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Amazon-Beauty-shaped synthetic 768-d, {args.levels}x{args.codes} codebooks, batch {args.batch}/GPU, "
                                   f"ROTATION_TRICK, {'tagged (projector+InfoNCE+predictor+focal/mixup: the step both h-configs run)' if args.tagged else 'untagged core'}"
                                   f" train step = fwd+bwd+{'RCCL all-reduce+' if world > 1 else ''}AdamW(cosine)",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "hip_graph": bool(use_graph),
                       "batch_formation": "every timed step gathers its batch from the resident item tables into the step's input buffers "
                                          "(ResidentItemData.gather_into: one hidvae_gather_rows launch, inside the timed region)",
                       "graph_queues": os.environ.get("DEBUG_HIP_FORCE_GRAPH_QUEUES"),
                       "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "4 (runtime default)"),
                       "collectives_in_graph": bool(getattr(info["stepper"], "in_graph", False)),
                       "dp_comm": (None if info["stepper"].dp is None else
                                   "own RCCL communicator (hidvae_amd.rccl)" if getattr(info["stepper"].dp, "_comm", None) else "torch.distributed")},
            "roofline": {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")} | {"kernel": roof["kernel"], "us_per_launch": roof["us"]}
                        | {k: roof[k] for k in ("share_of_step", "us_warm", "frac_warm", "largest_launch", "whole_step") if k in roof},
            "windows_ms_per_step": info["windows_ms_per_step"],
            "kernels": ks, "final_loss": final_loss,
        }
        if in_step is not None:
            line["in_step"] = in_step
        if other_extra is not None:
            line["untagged_core_step" if args.tagged else "tagged_step"] = other_extra
        if large_extra is not None:
            line["large_batch_step"] = large_extra
        line.update(config_extras)
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        degraded = bool(args.graph) and not use_graph  # the step could not be captured and ran eagerly (host-bound, ~6x slower)
        if degraded:
            line["degraded"] = True
        print(json.dumps(line), flush=True)
    if dist is not None:
        if world > 1:
            dist.barrier()  # the other ranks wait for rank 0's kernel timings instead of tearing the group down under it
        dist.destroy_process_group()
    if bool(args.graph) and not use_graph:
        print("[bench] the step was asked to replay from a HIP graph and could not be captured: the line above is marked degraded", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
