import sys, time, types
sys.path.insert(0, '.')
import torch, hidvae_amd
from hidvae_amd.modules.tokenizer.h_semids import HSemanticIdTokenizer
torch.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
tok = HSemanticIdTokenizer(input_dim=768, output_dim=32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3, n_cat_feats=0,
                           hrqvae_weights_path=None, hrqvae_codebook_normalize=True, tag_class_counts=[38, 168, 348], tag_embed_dim=768)
tok.hrq_vae.cuda().eval()
with torch.no_grad():
    for i, layer in enumerate(tok.hrq_vae.layers):
        layer.embedding.weight.copy_((torch.rand_like(layer.embedding.weight) * 2 - 1) * (1.0 if i == 0 else 0.35 * 0.5 ** i))
        layer.kmeans_initted = True
x = torch.nn.functional.normalize(torch.randn(N, 768, device="cuda"), dim=-1)
for _ in range(2):
    tok.reset(); ids = tok.precompute_corpus_ids(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    tok.reset(); ids = tok.precompute_corpus_ids(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"precompute_corpus_ids: {N} items in {dt*1e3:.2f} ms = {N/dt/1e6:.2f} M items/s; ids {tuple(ids.shape)}; distinct tuples {torch.unique(ids, dim=0).shape[0]}")
q = ids[torch.randint(0, N, (4096,), device='cuda')][:, :2]
torch.cuda.synchronize(); t0 = time.perf_counter()
hit = tok.exists_prefix(q)
torch.cuda.synchronize(); print("exists_prefix 4096 queries (first call builds the sorted index): %.2f ms, hits %d" % ((time.perf_counter() - t0) * 1e3, int(hit.sum())))
t0 = time.perf_counter(); hit = tok.exists_prefix(q); torch.cuda.synchronize(); print("exists_prefix again: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
