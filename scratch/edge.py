import sys, types
sys.path.insert(0, '.')
import torch, hidvae_amd
from hidvae_amd.modules.h_rqvae import HRqVae
from hidvae_amd.modules.quantize import QuantizeForwardMode as M
torch.manual_seed(0)
def run(B, hidden, K, L, mode=M.ROTATION_TRICK, tagged=False, norm=True):
    m = HRqVae(input_dim=768, embed_dim=32, hidden_dims=hidden, codebook_size=K, codebook_kmeans_init=False, codebook_normalize=norm,
               codebook_mode=mode, n_layers=L, commitment_weight=0.4, n_cat_features=0, tag_class_counts=[7, 9, 11, 5, 6, 7, 8, 9][:L]).cuda().train()
    b = types.SimpleNamespace(x=torch.nn.functional.normalize(torch.randn(B, 768), dim=-1).cuda())
    if tagged:
        b.tags_emb = torch.randn(B, L, 768).cuda(); b.tags_indices = torch.randint(0, 5, (B, L)).cuda()
    fused = m._bottleneck_ok(b.x)
    out = m(b, gumbel_t=0.2)
    out.loss.backward()
    gn = sum(float(p.grad.norm()) for p in m.parameters() if p.grad is not None)
    print(f"B={B} hidden={hidden} K={K} L={L} mode={mode.name} tagged={tagged}: fused={fused} loss={float(out.loss):.5f} gradsum={gn:.4f} p_unique={float(out.p_unique_ids):.3f}")
run(1, [512, 256, 128], 256, 3)
run(1, [512, 256, 128], 256, 3, tagged=True)
run(3, [512, 256, 128], 100, 3)
run(17, [512, 256, 128], 256, 1)
run(40, [512, 256, 128], 64, 8)
run(33, [256, 128], 256, 3)
run(33, [128], 256, 3)
run(33, [512, 256, 100], 256, 3)
run(5000, [512, 256, 128], 256, 3)
run(64, [512, 256, 128], 256, 3, mode=M.STE, norm=False)
run(64, [512, 256, 128], 256, 3, mode=M.GUMBEL_SOFTMAX)
