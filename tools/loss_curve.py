"""Loss curve of the HIP path (bf16 GEMMs, fp32 master weights / optimizer) against the fp32 oracle over many
optimizer steps on the tiny config, same seeded batches, dropout off (the oracle — like the reference's eval of
parity — has no dropout stream to share).  Writes a CSV and prints the summary the north star asks for
("loss curve within 1e-3 of reference over 1k steps").  Three curves on the same batches: the HIP path, the fp32 oracle, and
the oracle under the reference's OWN precision policy (`O.autocast_bf16()`: HF Trainer bf16=True autocast around compute_loss,
bf16 LLM weights) — the distance between the last two is what the reference itself shows between an autocast and an fp32
run, i.e. the floor any bf16 implementation of this step sits on.
  python tools/loss_curve.py [--steps 1000] [--qwen3] [--deep] [--out profiles/r02_loss_curve_tiny.csv]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("desta2.5-audio_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--qwen3", action="store_true")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--pool", type=int, default=16, help="distinct batches, cycled")
    ap.add_argument("--out", default="")
    ap.add_argument("--deep", action="store_true", help="tiny width at the reference's depth (32 / 6 / 32|36 layers)")
    ap.add_argument("--orca", action="store_true", help="ORCA hybrid (global + local tokens in the gated cross-attention, the shipped configs' switches): "
                                                        "curve of the trainer's TOTAL loss = LM + the three ORCA terms")
    a = ap.parse_args()
    torch.set_num_threads(min(8, torch.get_num_threads()))      # tiny CPU ops: more threads only add overhead
    import desta_oracle as O
    from helpers import cfg_from_dims
    from desta.models.modeling_desta25 import DeSTA25AudioModel
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments
    d = O.deep_dims(a.qwen3) if a.deep else O.tiny_dims(a.qwen3)
    if a.orca:
        import orca_oracle as R
        kg, ntr = 8, 3
        o = R.OrcaDims(global_num_tokens=kg, local_downsample=4, local_kernel_size=5, global_cross_attn=True, ortho_diversity_weight=0.05,
                       ortho_weight_qformer_local=0.05, align_weight_local=0.05)
        w = R.init_weights(d, o, seed=21)
        d.prompt_size = kg + ntr                                          # placeholders per audio: global tokens + a 3-token transcription
        cfg = cfg_from_dims(d, dropout=0.0, connector_mode="orca_hybrid", orca_enabled=True, orca_global_num_tokens=kg, orca_local_downsample=4,
                            orca_local_kernel_size=5, orca_global_cross_attn=True, orca_ortho_diversity_weight=0.05,
                            orca_ortho_weight_qformer_local=0.05, orca_align_weight_local=0.05)
        names = R.trainable_names(d, o)
    else:
        w = O.init_weights(d, seed=21)
        cfg = cfg_from_dims(d, dropout=0.0)
        names = O.trainable_names(d)
    model = DeSTA25AudioModel(cfg, weights=w)
    w = {k: v.clone() for k, v in w.items()}
    tr = DeSTA25Trainer(model, args=TrainingArguments(learning_rate=a.lr, warmup_steps=a.warmup, max_steps=a.steps, logging_steps=10 ** 9))
    w_ac = {k: v.clone() for k, v in w.items()}
    st, st_ac = O.adafactor_init([w[n] for n in names]), O.adafactor_init([w[n] for n in names])
    pool = [O.synthetic_batch(d, B=2, S_ctx=5, S_tgt=24, seed=500 + i, pad=[0, i % 3]) for i in range(a.pool)]
    if a.orca:
        g = torch.Generator().manual_seed(77)
        for b in pool:
            b["batch_transcription_ids"] = [torch.randint(3, d.vocab, (1, ntr), generator=g) for _ in range(2)]

    def orca_step(wd_, st_, batch, lr, autocast):
        """forward -> LM + ORCA losses -> backward -> clip 1.0 -> Adafactor, the trainer's order (desta_trainer.py:56-92)."""
        for n in names:
            wd_[n].requires_grad_(True)
            wd_[n].grad = None
        if autocast:
            with O.autocast_bf16():
                loss, _, losses = R.model_forward(wd_, d, o, batch, training=True)
                tot = R.total_loss(loss, losses)
        else:
            loss, _, losses = R.model_forward(wd_, d, o, batch, training=True)
            tot = R.total_loss(loss, losses)
        tot.backward()
        grads = [wd_[n].grad.detach().clone() for n in names]
        params = [wd_[n].detach() for n in names]
        O.clip_grad_norm(grads, 1.0)
        with torch.no_grad():
            O.adafactor_step(params, grads, st_, lr, [0.01 if m else 0.0 for m in O.decay_mask(names)])
        for n, p_ in zip(names, params):
            wd_[n] = p_.detach()
        return tot.detach()
    t0 = time.time()
    hip = tr.train([pool[i % a.pool] for i in range(a.steps)])
    t_hip = time.time() - t0
    t0 = time.time()
    ref, rac = [], []
    for i in range(a.steps):
        lr = O.linear_warmup_lr(i, a.lr, a.warmup, a.steps)
        if a.orca:
            lo, la = orca_step(w, st, pool[i % a.pool], lr, False), orca_step(w_ac, st_ac, pool[i % a.pool], lr, True)
        else:
            lo, _, _, _ = O.train_step(w, d, pool[i % a.pool], st, lr)
            la, _, _, _ = O.train_step(w_ac, d, pool[i % a.pool], st_ac, lr, autocast=True)
        ref.append(float(lo))
        rac.append(float(la))
        if i % 25 == 0:
            print(f"[oracle] step {i}: fp32 {ref[-1]:.4f} autocast {rac[-1]:.4f} (HIP {hip[i]:.4f})  {time.time() - t0:.0f}s", flush=True)
    t_ref = time.time() - t0
    k = max(1, a.steps // 10)

    def summary(x, y, tag):
        diff = [abs(p - q) for p, q in zip(x, y)]
        print(f"|{tag}|: mean {sum(diff) / len(diff):.2e}, max {max(diff):.2e}, last-{k} mean {sum(diff[-k:]) / k:.2e}, "
              f"|mean of last-{k} losses| {abs(sum(x[-k:]) - sum(y[-k:])) / k:.2e}")
    print(f"steps {a.steps}: fp32 oracle loss {ref[0]:.4f} -> {sum(ref[-k:]) / k:.4f} (mean of last {k}), autocast oracle -> {sum(rac[-k:]) / k:.4f}, "
          f"HIP {hip[0]:.4f} -> {sum(hip[-k:]) / k:.4f}; HIP {t_hip:.1f}s, oracles {t_ref:.1f}s")
    summary(hip, ref, "HIP - fp32 oracle")
    summary(rac, ref, "autocast oracle - fp32 oracle (the reference's own policy)")
    summary(hip, rac, "HIP - autocast oracle")
    if a.out:
        with open(a.out, "w") as f:
            f.write("step,loss_hip_bf16,loss_oracle_fp32,loss_oracle_autocast_bf16\n")
            for i, (x, y, z) in enumerate(zip(hip, ref, rac)):
                f.write(f"{i},{x:.6f},{y:.6f},{z:.6f}\n")


if __name__ == "__main__":
    main()
