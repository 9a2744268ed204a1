import sys, time, tempfile
sys.path.insert(0, '.')
import torch, hidvae_amd
from hidvae_amd.modules.quantize import QuantizeForwardMode
from hidvae_amd.train_hidvae import train
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
t0 = time.time()
with tempfile.TemporaryDirectory() as d:
    model, series = train(iterations=iters, batch_size=1024, learning_rate=2.8e-4, weight_decay=0.015, dataset="synthetic:20000", save_dir_root=d + "/",
        use_kmeans_init=True, do_eval=True, gradient_accumulate_every=1, eval_every=iters, commitment_weight=0.4, tag_alignment_weight=0.15,
        tag_prediction_weight=0.55, vae_n_cat_feats=0, vae_input_dim=768, vae_embed_dim=32, vae_hidden_dims=[512, 256, 128],
        vae_codebook_size=256, vae_codebook_normalize=True, vae_codebook_mode=QuantizeForwardMode.ROTATION_TRICK, vae_n_layers=3,
        tag_class_counts=[38, 168, 348], use_focal_loss=True, focal_loss_gamma_base=2.7, focal_loss_alpha_base=0.24, rare_tag_threshold=5,
        dropout_rate=0.4, lr_scheduler_T_max=400000, lr_scheduler_eta_min=7e-8, sem_id_uniqueness_weight=1.5, sem_id_uniqueness_margin=0.0,
        id_repetition_threshold=0.06, log_every=max(1, iters // 10))
dt = time.time() - t0
print("wall s", round(dt, 1), "ms/iter incl. startup+eval", round(dt / iters * 1e3, 3))
for it, row in zip(series["iter"], series["loss"]):
    print(it, [round(v, 4) for v in row])
print({k: v for k, v in series["eval"][-1].items() if not isinstance(v, list)})
print("mem MB", torch.cuda.max_memory_allocated() / 1e6)
