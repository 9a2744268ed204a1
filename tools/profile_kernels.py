#!/usr/bin/env python3
"""Launch the kernels quoted in bench.py's `roofline`/`kernels` a few times each, for rocprofv3 passes:
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/profile_kernels.py
    rocprofv3 --kernel-trace --pmc FETCH_SIZE  --output-format csv -d OUT -- python3 tools/profile_kernels.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE  --output-format csv -d OUT -- python3 tools/profile_kernels.py
(counters in their own passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hidvae_amd  # noqa: E402,F401
from hidvae_amd import _C  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
B, L, K = 1024, 3, 256
x = torch.randn(B, 768, device=dev)
w0 = torch.randn(512, 768, device=dev) * 0.03
o0, a0 = torch.empty(B, 512, device=dev), torch.empty(B, 512, device=dev)
g = torch.randn(B, 512, device=dev)
gw = torch.empty(512, 768, device=dev)
tabs = [(torch.rand(K, 32, device=dev) * 2 - 1) * (1.0 if i == 0 else 0.35 * 0.5 ** i) for i in range(L)]
cb, cc = _C.codebook_prepare(tabs, [i == 0 for i in range(L)])
y_small = torch.randn(B, 32, device=dev)
y_big = torch.randn(1 << 20, 32, device=dev)
w1 = torch.randn(256, 512, device=dev) * 0.03
pre = torch.randn(B, 512, device=dev)
g1 = torch.randn(B, 256, device=dev)
for _ in range(10):
    _C.gemm(_C.GEMM_NT, x, w0, out=o0, epilogue=_C.EPI_SILU, aux=a0)
    _C.linear_bwd(g, x, None, False, dW=gw, bias=True)  # encoder layer 0 backward as the step launches it: weight gradient + bias, no dX
    _C.linear_bwd(g1, o0, w1, True, _C.EPI_DSILU, pre)  # the paired dW + dX launch of encoder layer 1
    _C.rq_forward(y_small, cb, cc, True, 3, True, 0.4)
# round 2: the launch family that leads the step (hidvae_linear_bwd on the decoder's last layer), the streamed
# code-split RQ kernel of config 5 (4 x 1024 codes, B = 4096)
wd3 = torch.randn(768, 512, device=dev) * 0.03
gd, xd, pd = torch.randn(B, 768, device=dev), torch.randn(B, 512, device=dev), torch.randn(B, 512, device=dev)
tabs5 = [(torch.rand(1024, 32, device=dev) * 2 - 1) * (1.0 if i == 0 else 0.35 * 0.5 ** i) for i in range(4)]
cb5, cc5 = _C.codebook_prepare(tabs5, [i == 0 for i in range(4)])
y5 = torch.randn(4096, 32, device=dev)
x8 = torch.randn(8192, 768, device=dev)
o8, a8 = torch.empty(8192, 512, device=dev), torch.empty(8192, 512, device=dev)
g8 = torch.randn(8192, 256, device=dev)
for _ in range(10):
    _C.linear_bwd(gd, xd, wd3, True, _C.EPI_DSILU, pd)
    _C.rq_forward(y5, cb5, cc5, True, 3, True, 0.4)
    _C.gemm(_C.GEMM_NT, x8, w0, out=o8, epilogue=_C.EPI_SILU, aux=a8)       # fp32 LDS-tiled at B = 8192
xb = torch.randn(1 << 16, 768, device=dev)
ob, ab = torch.empty(1 << 16, 512, device=dev), torch.empty(1 << 16, 512, device=dev)
# round 3: the tag heads' level-2 shapes (the widest Linear backward, LayerNorm both ways, the gate) and the ids-only corpus search
gt, xt, wt, yt = torch.randn(B, 691, device=dev), torch.randn(B, 768, device=dev), torch.randn(691, 768, device=dev) * 0.03, torch.rand(B, 768, device=dev)
gam, bet = torch.ones(768, device=dev), torch.zeros(768, device=dev)
xc = torch.randn(B, 96, device=dev)
gW = [torch.randn(24, 96, device=dev) * 0.1, torch.randn(24, device=dev), torch.randn(48, 24, device=dev) * 0.1, torch.randn(48, device=dev),
      torch.randn(96, 48, device=dev) * 0.1, torch.randn(96, device=dev)]
for _ in range(10):
    _C.linear_bwd(gt, xt, wt, True, _C.EPI_DRELU, yt, bias=True, dx_scale=1.6)   # 1024 x 691 x 768: the heads' widest backward launch
    yl, ml, rl = _C.layernorm_fwd(xt, gam, bet, 1e-5, True, None, 1.0, None)
    _C.layernorm_bwd_partial(xt, xt, gam, bet, ml, rl, True, yl, 1.0, 0.0)
    hh, sv = _C.gate_fwd(xc, *gW, True)
    _C.gate_bwd(hh, xc, gW[0], gW[2], gW[4], True, sv)
# round 4: every Linear backward from 0.2 GFLOP on runs on the LDS-DMA ring kernel (gemm_ring.hip); the heads' widest layer at config 3's batch
B2 = 2048
gt2, xt2, yt2 = torch.randn(B2, 691, device=dev), torch.randn(B2, 768, device=dev), torch.rand(B2, 768, device=dev)
for _ in range(10):
    _C.linear_bwd(gt2, xt2, wt, True, _C.EPI_DRELU, yt2, bias=True, dx_scale=1.6)   # 2048 x 691 x 768: gemm_ring_bwd_kernel
# the loader's batch formation: rows of the resident item tables picked by the batch's ids, one launch (hidvae_gather_rows)
n_items = 8 * B
gx, gte, gti = torch.randn(n_items, 768, device=dev), torch.randn(n_items, L, 768, device=dev), torch.randint(0, 38, (n_items, L), device=dev)
gidx = torch.randperm(n_items, device=dev)[:B].contiguous()
gout = [torch.empty(B, 768, device=dev), torch.empty(B, L, 768, device=dev), torch.empty(B, L, dtype=torch.int64, device=dev)]
for _ in range(10):
    _C.gather_rows(gidx, [gx, gte, gti], gout)
for _ in range(5):
    _C.rq_forward(y_big, cb, cc, True, 3, True, 0.4)
    _C.rq_ids(y_big, cb, cc, True)                                     # the tokenizer's corpus pass: only the ids leave the launch
    _C.gemm(_C.GEMM_NT, xb, w0, out=ob, epilogue=_C.EPI_SILU, aux=ab)  # the LDS-tiled kernel (throughput regime)
torch.cuda.synchronize()
print("done")
