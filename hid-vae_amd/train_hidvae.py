"""Training driver for the HiD-VAE tokenizer on MI355X: the same gin-configurable `train()` surface as the reference
(train_hidvae.py:66-135, 61 keyword arguments) over the HIP model, the fused AdamW and RCCL data parallelism.

What it keeps from the reference loop (train_hidvae.py:687-1177): k-means warm-up forward at iteration 0, gradient
accumulation (losses/ga summed, ONE backward), AdamW param groups with layer-specific lr / weight decay, cosine (or step)
schedule stepped once per iteration, 1000-iteration sliding-window logs every 100 iterations, evaluation every
`eval_every` with the 5x noisy test-time-augmentation tag predictions, corpus semantic-id diversity statistics through
HSemanticIdTokenizer, and a checkpoint only when eval accuracy > 0.60 and the id repetition rate is below the threshold,
in the reference's dict layout.

What it does differently on purpose: items live in HBM and batches are drawn on the device by a per-rank seeded sampler
(SURVEY Q9); no per-iteration host synchronisation (scalars stay on the device between log lines); `accelerate` is
replaced by torch.distributed + one flat-gradient all-reduce; the matplotlib plots at the end are replaced by a JSON dump
of the same series."""
import json
import logging
import os
import sys
from collections import deque
from datetime import datetime

import torch

from . import gin_compat as gin
from .data.items import RandomBatches, ResidentItemData
from .modules.h_rqvae import HRqVae
from .modules.quantize import QuantizeForwardMode
from .modules.tokenizer.h_semids import HSemanticIdTokenizer
from .optim import HidvaeAdamW
from .parallel import DataParallel
from .step import GraphedTrainStep


def calculate_repetition_rate(item_ids):
    """1 - distinct/total over id tuples (reference train_hidvae.py:39-63)."""
    if item_ids is None or item_ids.nelement() == 0:
        return 0.0, 0, 0
    n_unique, total = torch.unique(item_ids, dim=0).shape[0], item_ids.shape[0]
    return 1.0 - n_unique / total, n_unique, total


def remap_rare_tags(train_ti, eval_ti, class_counts, n_layers, threshold):
    """Classes seen fewer than `threshold` times collapse into one trailing 'special' class; the others keep their order
    (reference train_hidvae.py:359-486).  Returns (new_class_counts, rare_ids_by_level, full_counts_by_level)."""
    new_counts, rare, full = [], {}, {}
    for i in range(n_layers):
        col = train_ti[:, i]
        valid = col[col >= 0]
        if valid.numel() == 0:
            new_counts.append(class_counts[i])
            continue
        counts = torch.bincount(valid, minlength=class_counts[i])[: class_counts[i]]
        rare_mask = (counts > 0) & (counts < threshold)
        rare[i] = torch.nonzero(rare_mask).squeeze(-1)
        full[i] = counts
        new_counts.append(int(((counts >= threshold) | (counts == 0)).sum()) + 1)
        if rare[i].numel() == 0:
            continue
        keep = ~rare_mask
        mapping = torch.cumsum(keep.long(), 0) - 1
        mapping[rare_mask] = new_counts[i] - 1
        for ti in (train_ti, eval_ti):
            if ti is not None:
                c = ti[:, i]
                ok = c >= 0
                c[ok] = mapping[c[ok]]
    return new_counts, rare, full


def _load_items(dataset, dataset_folder, device, n_layers, class_counts, tag_embed_dim, input_dim):
    """-> (train, eval, all) ResidentItemData.  `dataset` may be a ResidentItemData, a dict with tensors, or a path to a
    .pt file with {x, tags_emb, tags_indices, is_train}; 'synthetic:N' builds N random items."""
    if isinstance(dataset, ResidentItemData):
        full, is_train = dataset, None
    elif isinstance(dataset, dict):
        full, is_train = ResidentItemData(dataset["x"], dataset.get("tags_emb"), dataset.get("tags_indices"), device=device), dataset.get("is_train")
    elif isinstance(dataset, str) and dataset.startswith("synthetic:"):
        full, is_train = ResidentItemData.synthetic(int(dataset.split(":")[1]), input_dim, n_layers, tuple(class_counts), tag_embed_dim,
                                                    device=device), None
    else:
        name = dataset.name.lower() if hasattr(dataset, "name") else "items"
        path = dataset if isinstance(dataset, str) and os.path.exists(str(dataset)) else os.path.join(dataset_folder, f"{name}.pt")
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"no item tensors at {path}: the reference's dataset pipeline (gdrive download + sentence-T5, torch_geometric) is out "
                "of scope; pass a .pt with x / tags_emb / tags_indices / is_train, a dict, a ResidentItemData or 'synthetic:N'")
        full, is_train = ResidentItemData.from_file(path, device)
    if is_train is None:
        g = torch.Generator().manual_seed(0)
        is_train = (torch.rand(len(full), generator=g) < 0.95).to(full.x.device)
    is_train = is_train.to(full.x.device)
    return full.subset(is_train), full.subset(~is_train), full


@gin.configurable
def train(iterations=50000, batch_size=64, learning_rate=0.0001, weight_decay=0.01, dataset_folder="dataset/ml-1m", dataset=None,
          pretrained_hrqvae_path=None, save_dir_root="out/", use_kmeans_init=True, split_batches=True, amp=False, do_eval=True,
          force_dataset_process=False, mixed_precision_type="fp16", gradient_accumulate_every=1, save_model_every=1000000,
          eval_every=50000, commitment_weight=0.25, tag_alignment_weight=0.5, tag_prediction_weight=0.5, vae_n_cat_feats=18,
          vae_input_dim=18, vae_embed_dim=16, vae_hidden_dims=[18, 18], vae_codebook_size=32, vae_codebook_normalize=False,
          vae_codebook_mode=QuantizeForwardMode.GUMBEL_SOFTMAX, vae_sim_vq=False, vae_n_layers=3, dataset_split="beauty",
          tag_class_counts=None, tag_embed_dim=768, use_focal_loss=False, focal_loss_gamma_base=2.0, focal_loss_alpha_base=0.25,
          rare_tag_threshold=30, dropout_rate=0.3, use_batch_norm=True, alignment_temperature=0.1, predictor_weight_decay=0.01,
          layer_specific_lr=True, use_label_smoothing=True, label_smoothing_alpha=0.1, use_mixup=True, mixup_alpha=0.2,
          eval_tta=True, eval_temperature=0.8, ensemble_predictions=True, use_lr_scheduler=True, lr_scheduler_type="cosine",
          lr_scheduler_T_max=400000, lr_scheduler_eta_min=1e-6, lr_scheduler_step_size=100000, lr_scheduler_gamma=0.5,
          lr_scheduler_factor=0.5, lr_scheduler_patience=10, sem_id_uniqueness_weight=0.5, sem_id_uniqueness_margin=0.5,
          id_repetition_threshold=0.03, use_concatenated_ids=False, use_interleaved_ids=False, seed=0, log_every=100,
          use_hip_graph=True):
    if amp:
        raise NotImplementedError("amp is False in every reference config; the HIP path computes in fp32")
    if lr_scheduler_type not in ("cosine", "step"):
        raise ValueError(f"unsupported lr_scheduler_type {lr_scheduler_type}")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 and not torch.distributed.is_initialized():
        # "nccl" IS RCCL on ROCm; HIDVAE_DIST_BACKEND=gloo lets several ranks share one GPU (tests), where RCCL refuses duplicates
        backend = os.environ.get("HIDVAE_DIST_BACKEND", "nccl")
        torch.distributed.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
    main_proc = rank == 0

    save_dir = os.path.join(save_dir_root, datetime.now().strftime("%Y%m%d_%H%M%S"))
    log = logging.getLogger("hidvae.train")
    log.setLevel(logging.INFO if main_proc else logging.WARNING)
    if main_proc and not log.handlers:
        os.makedirs(save_dir, exist_ok=True)
        for h in (logging.StreamHandler(sys.stdout), logging.FileHandler(os.path.join(save_dir, "train.log"))):
            h.setFormatter(logging.Formatter("%(asctime)s %(message)s"))
            log.addHandler(h)

    L = vae_n_layers
    tag_class_counts = list(tag_class_counts) if tag_class_counts is not None else [10, 100, 1000][:L]
    train_set, eval_set, all_set = _load_items(dataset, dataset_folder, device, L, tag_class_counts, tag_embed_dim, vae_input_dim)
    has_tags = train_set.has_tags
    if has_tags:  # keep the first n_layers tag levels (reference train_hidvae.py:252-267)
        for ds in (train_set, eval_set, all_set):
            ds.tags_emb, ds.tags_indices = ds.tags_emb[:, :L].contiguous(), ds.tags_indices[:, :L].contiguous()
    focal_params = None
    if use_focal_loss:
        focal_params = {}
        for i in range(L):
            focal_params[f"gamma_{i}"], focal_params[f"alpha_{i}"] = focal_loss_gamma_base, focal_loss_alpha_base
    class_counts_dict = {}
    if has_tags and use_focal_loss:
        tag_class_counts, rare, class_counts_dict = remap_rare_tags(train_set.tags_indices, eval_set.tags_indices if do_eval else None,
                                                                    tag_class_counts, L, rare_tag_threshold)
        if main_proc:
            rare_path = os.path.join(save_dir_root + "special_tags_files", "rare_tags.pt")
            os.makedirs(os.path.dirname(rare_path), exist_ok=True)
            torch.save({k: v.cpu() for k, v in rare.items()}, rare_path)
            log.info(f"Updated number of tag classes: {tag_class_counts}; rare tag ids saved to {rare_path}")

    torch.manual_seed(seed)
    import numpy as np
    np.random.seed(seed % (2 ** 32))  # the k-means start draws np.random.choice (init/kmeans.py:40 of the reference): make a run replayable
    model = HRqVae(input_dim=vae_input_dim, embed_dim=vae_embed_dim, hidden_dims=vae_hidden_dims, codebook_size=vae_codebook_size,
                   codebook_kmeans_init=use_kmeans_init and pretrained_hrqvae_path is None, codebook_normalize=vae_codebook_normalize,
                   codebook_sim_vq=vae_sim_vq, codebook_mode=vae_codebook_mode, n_layers=L, n_cat_features=vae_n_cat_feats,
                   commitment_weight=commitment_weight, tag_alignment_weight=tag_alignment_weight,
                   tag_prediction_weight=tag_prediction_weight, tag_class_counts=tag_class_counts, tag_embed_dim=tag_embed_dim,
                   use_focal_loss=use_focal_loss, focal_loss_params=focal_params, dropout_rate=dropout_rate, use_batch_norm=use_batch_norm,
                   alignment_temperature=alignment_temperature, sem_id_uniqueness_weight=sem_id_uniqueness_weight,
                   sem_id_uniqueness_margin=sem_id_uniqueness_margin).to(device)
    if class_counts_dict:
        model.update_class_counts({k: v.to(device) for k, v in class_counts_dict.items()})
    tpl = model.tag_prediction_loss
    tpl.use_label_smoothing, tpl.label_smoothing_alpha, tpl.use_mixup, tpl.mixup_alpha = use_label_smoothing, label_smoothing_alpha, use_mixup, mixup_alpha

    if layer_specific_lr:  # reference train_hidvae.py:537-561
        groups = [{"params": list(model.encoder.parameters()) + list(model.decoder.parameters()), "lr": learning_rate, "weight_decay": weight_decay},
                  {"params": [p for layer in model.layers for p in layer.parameters()], "lr": learning_rate, "weight_decay": weight_decay}]
        for i in range(L):
            plr = learning_rate * (1 + i * 0.1)
            pwd = predictor_weight_decay / (1 + i * 0.2) if predictor_weight_decay > 0 else predictor_weight_decay
            groups.append({"params": list(model.tag_predictors[i].parameters()), "lr": plr, "weight_decay": pwd})
            groups.append({"params": list(model.tag_projectors[i].parameters()), "lr": plr, "weight_decay": pwd})
    else:
        groups = [{"params": list(model.parameters()), "lr": learning_rate, "weight_decay": weight_decay}]
    start_iter, opt_state = 0, None
    if pretrained_hrqvae_path is not None:
        from .checkpoint import load_checkpoint
        model.load_pretrained(pretrained_hrqvae_path)
        state = load_checkpoint(pretrained_hrqvae_path, map_location=device)
        start_iter, opt_state = state["iter"] + 1, state.get("optimizer")
    cosine = (lr_scheduler_T_max, lr_scheduler_eta_min) if (use_lr_scheduler and lr_scheduler_type == "cosine") else None
    step_lr = (lr_scheduler_step_size, lr_scheduler_gamma) if (use_lr_scheduler and lr_scheduler_type == "step") else None
    opt = HidvaeAdamW(groups, cosine=cosine, step_lr=step_lr, start_step=start_iter, flat_grads=world > 1,
                      first_bucket=model.dp_first_bucket(batch_size) if world > 1 else None).prepare()
    if pretrained_hrqvae_path is not None:  # reference train_hidvae.py:625: optimizer.load_state_dict(state["optimizer"])
        restored = False
        if opt_state is not None:
            try:
                restored = opt.load_state_dict(opt_state)
            except (ValueError, KeyError) as e:
                log.warning(f"optimizer state of {pretrained_hrqvae_path} does not fit this model ({e})")
        if not restored:
            opt.restart_without_state(start_iter)
            log.warning("resuming WITHOUT optimizer state: AdamW moments start at zero and its bias correction restarts from step 0; "
                        f"the learning-rate schedule continues from iteration {start_iter}")
    dp = DataParallel(model, opt.grad_buffer) if world > 1 else None
    if dp is not None:
        dp.broadcast_parameters(0)

    tokenizer = HSemanticIdTokenizer(input_dim=vae_input_dim, output_dim=vae_embed_dim, hidden_dims=vae_hidden_dims,
                                     codebook_size=vae_codebook_size, n_layers=L, n_cat_feats=vae_n_cat_feats,
                                     hrqvae_weights_path=None, hrqvae_codebook_normalize=vae_codebook_normalize, hrqvae_sim_vq=vae_sim_vq,
                                     tag_alignment_weight=tag_alignment_weight, tag_prediction_weight=tag_prediction_weight,
                                     tag_class_counts=tag_class_counts, tag_embed_dim=tag_embed_dim,
                                     use_concatenated_ids=use_concatenated_ids, use_interleaved_ids=use_interleaved_ids,
                                     commitment_weight=commitment_weight)
    tokenizer.hrq_vae = model

    sampler = RandomBatches(train_set, batch_size, seed=(dp.shard_seed(seed) if dp else seed))
    window = deque(maxlen=1000)  # per-iteration device rows; read back only when a log line is due
    stepper = None
    series = {"iter": [], "loss": [], "eval": []}
    t = 0.2
    one = torch.ones((), device=device)
    ga = gradient_accumulate_every
    for it in range(start_iter, start_iter + 1 + iterations):
        model.train()
        if it == 0 and use_kmeans_init:
            # the reference's warm-up: one full training-mode forward on the first min(20000, N) items whose only wanted effect
            # is the k-means codebook initialisation (train_hidvae.py:692-696; BatchNorm buffers see this batch too)
            # (no autograd tape: nothing is differentiated here, and at 20,000 tagged items the saved activations of the three heads
            #  would be gigabytes; the InfoNCE term itself never materialises its B x B matrix from 4096 items on)
            with torch.no_grad():
                model(train_set[torch.arange(min(20000, len(train_set)), device=device)], t)
            if dp is not None:  # rank 0's codebooks win (the reference lets them diverge)
                dp.broadcast_codebooks([layer.embedding.weight for layer in model.layers], 0)
            log.info("K-means initialization complete")
        # (once the step object exists the batches are gathered straight into its input buffers: no per-step copy)
        micro = [sampler.next(out=stepper.input_buffers(j) if stepper is not None else None) for j in range(ga)]
        if stepper is None:  # built on the first regular step: its static buffers take the batch shapes
            stepper = GraphedTrainStep(model, opt, micro, dp=dp, gumbel_t=t, enabled=use_hip_graph)
        row = stepper(micro)  # device [6]: total loss, mean recon, mean rqvae, tag align, tag pred, tag accuracy
        window.append(row.clone())
        if it % log_every == 0 and main_proc:
            m = torch.stack(list(window)).mean(0).tolist()
            series["iter"].append(it)
            series["loss"].append(m)
            log.info(f"Iter {it} - loss: {m[0]:.4f}, rl: {m[1]:.4f}, vl: {m[2]:.4f}, tal: {m[3]:.4f}, tpl: {m[4]:.4f}, acc: {m[5]:.4f}, "
                     f"lr: {opt.current_lr():.3e}")
        if do_eval and ((it + 1) % eval_every == 0 or it + 1 == iterations) and len(eval_set) > 0:
            ev = evaluate(model, eval_set, all_set, tokenizer, L, vae_codebook_size, t, eval_tta, eval_temperature, batch_size, log if main_proc else None)
            series["eval"].append({"iter": it + 1, **{k: (float(v) if not isinstance(v, list) else v) for k, v in ev.items()}})
            if main_proc and ev["eval_tag_pred_accuracy"] > 0.60 and ev["sem_id_repetition_rate"] < id_repetition_threshold:
                os.makedirs(save_dir, exist_ok=True)
                name = (f"hrqvae_model_ACC{ev['eval_tag_pred_accuracy']:.4f}_RQLOSS{ev['eval_rqvae_loss']:.4f}_"
                        f"DUPR{ev['sem_id_repetition_rate']:.4f}_{datetime.now().strftime('%Y%m%d_%H%M%S')}.pt")
                from .checkpoint import save_checkpoint
                save_checkpoint({"iter": it + 1, "model": model.state_dict(), "model_config": model.config, "optimizer": opt.state_dict(),
                                 "accuracy": ev["eval_tag_pred_accuracy"], "rqvae_loss": ev["eval_rqvae_loss"],
                                 "sem_id_repetition_rate": ev["sem_id_repetition_rate"]}, os.path.join(save_dir, name))
                log.info(f"Model saved to: {os.path.join(save_dir, name)}")
    if main_proc:
        os.makedirs(save_dir, exist_ok=True)
        with open(os.path.join(save_dir, "series.json"), "w") as f:
            json.dump(series, f)
    return model, series


@torch.no_grad()
def evaluate(model, eval_set, all_set, tokenizer, L, codebook_size, t, eval_tta, eval_temperature, batch_size, log=None):
    """Eval-mode losses over the eval items, TTA tag accuracy on up to 100 tagged samples, corpus id diversity
    (reference train_hidvae.py:810-1134)."""
    model.eval()
    sums, n = None, 0
    for lo in range(0, len(eval_set), batch_size):
        out = model(eval_set[torch.arange(lo, min(lo + batch_size, len(eval_set)), device=eval_set.x.device)], gumbel_t=t)
        row = torch.stack([out.loss, out.reconstruction_loss.mean(), out.rqvae_loss.mean(), out.tag_align_loss, out.tag_pred_loss,
                           out.tag_pred_accuracy])
        sums = row if sums is None else sums + row
        n += 1
    means = (sums / n).tolist()
    res = dict(zip(("eval_total_loss", "eval_reconstruction_loss", "eval_rqvae_loss", "eval_tag_align_loss", "eval_tag_pred_loss",
                    "eval_tag_pred_accuracy"), means))
    if eval_set.has_tags and eval_tta:
        valid = (eval_set.tags_indices >= 0).any(dim=1)
        pick = torch.nonzero(valid).squeeze(-1)[:100]
        if pick.numel() > 0:
            x = eval_set.x[pick]
            probs = [[] for _ in range(L)]
            for aug in range(5):  # original + 4 noisy copies, noise 0.02*aug (train_hidvae.py:874-917)
                xa = x if aug == 0 else x + torch.randn_like(x) * (0.02 * aug)
                r = model.encode(xa)
                embs = []
                for i, layer in enumerate(model.layers):
                    q = layer(r, temperature=0.001)
                    embs.append(q.embeddings)
                    logits = model.tag_predictors[i](torch.cat(embs, dim=-1)) / eval_temperature
                    probs[i].append(torch.softmax(logits, dim=-1))
                    r = r - q.embeddings
            truth = eval_set.tags_indices[pick]
            accs = []
            for i in range(L):
                pred = torch.stack(probs[i]).mean(0).argmax(-1)
                ok = truth[:, i] >= 0
                accs.append(float((pred[ok] == truth[ok, i]).float().mean()) if ok.any() else 0.0)
            res["tta_accuracy_by_layer"] = accs
    tokenizer.reset()
    corpus = tokenizer.precompute_corpus_ids(all_set.x)
    _, counts = torch.unique(corpus[:, L - 1], dim=0, return_counts=True)
    p = counts / corpus.shape[0]
    res["rqvae_entropy"] = float(-(p * torch.log(p)).sum())
    res["max_id_duplicates"] = float(corpus[:, -1].max() / corpus.shape[0])
    res["codebook_usage"] = [float(torch.unique(corpus[:, i]).numel() / codebook_size) for i in range(L)]
    rep, n_unique, total = calculate_repetition_rate(corpus[:, :L])
    res["sem_id_repetition_rate"] = rep
    if log is not None:
        log.info(f"Eval - " + ", ".join(f"{k}: {v:.4f}" for k, v in res.items() if isinstance(v, float))
                 + f" | usage {res['codebook_usage']} | unique ids {n_unique}/{total}")
    return res


def parse_config():
    """`python -m hidvae_amd.train_hidvae configs/h_rqvae_amazon.gin` (reference modules/utils.py:58-62)."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("config_path", type=str, help="Path to gin config file.")
    args = ap.parse_args()
    gin.parse_config_file(args.config_path, import_aliases={"data.tags_processed": "hidvae_amd.data.items", "modules.quantize": "hidvae_amd.modules.quantize"})


if __name__ == "__main__":
    parse_config()
    train()
