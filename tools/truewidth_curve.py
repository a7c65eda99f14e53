"""Loss curve at TRUE WIDTH over a few optimizer steps: the HIP path against the fp32 oracle on the same seeded batches
(whisper-large-v3 width with 4 encoder layers / 1500 frames, Q-Former 2L at d = 1280, Llama-3.1-8B width with 2 decoder layers;
B = 2, S = 168; dropout off) — the widths at which the one-pass cross-attention backward, the transposed d(K|V) and the bias sums
inside it run (the tiny configs of tools/loss_curve.py have 96 encoder frames and take the two-kernel path).  The oracle is
~15 s per step on the host cores.
  python tools/truewidth_curve.py [--steps 12] [--out profiles/r03_truewidth_curve.csv]"""
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
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--out", default="")
    ap.add_argument("--autocast", action="store_true", help="third curve: the oracle under the reference's own bf16 autocast policy (doubles the host time)")
    a = ap.parse_args()
    import desta_oracle as O
    from helpers import cfg_from_dims
    from desta.models.modeling_desta25 import DeSTA25AudioModel
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments
    torch.set_num_threads(max(1, min(64, len(os.sched_getaffinity(0)))))
    d = O.Dims(n_mels=128, enc_d=1280, enc_layers=4, enc_heads=20, enc_ffn=5120, enc_T=1500, taps=(0, 1, 2, 3), qf_layers=2, qf_inter=3072,
               prompt_size=64, llm_h=4096, llm_layers=2, llm_hq=32, llm_hkv=8, llm_hd=128, tie_embeddings=False, llm_inter=14336, vocab=128256,
               rms_eps=1e-5, rope_theta=500000.0, rope_llama3=(8.0, 1.0, 4.0, 8192), qk_norm=False)
    w = O.init_weights(d, seed=11)
    model = DeSTA25AudioModel(cfg_from_dims(d), weights=w)
    tr = DeSTA25Trainer(model, args=TrainingArguments(learning_rate=a.lr, warmup_steps=2, max_steps=a.steps, weight_decay=0.01, overlap_comm=False))
    pool = [O.synthetic_batch(d, B=2, S_ctx=24, S_tgt=80, seed=100 + i, pad=[0, 5]) for i in range(4)]
    names = O.trainable_names(d)
    st = O.adafactor_init([w[n] for n in names])
    wa = {k: (v.clone() if k in names else v) for k, v in w.items()} if a.autocast else None      # frozen tensors shared, trainable ones copied
    sta = O.adafactor_init([wa[n] for n in names]) if a.autocast else None
    rows = []
    t_h = t_o = 0.0
    for i in range(a.steps):
        b = pool[i % len(pool)]
        t0 = time.time()
        lh = float(tr.training_step(b))
        tr.wait_update()
        torch.cuda.synchronize()
        t_h += time.time() - t0
        t0 = time.time()
        lo = float(O.train_step(w, d, b, st, O.linear_warmup_lr(i, a.lr, 2, a.steps), weight_decay=0.01)[0])
        la = float(O.train_step(wa, d, b, sta, O.linear_warmup_lr(i, a.lr, 2, a.steps), weight_decay=0.01, autocast=True)[0]) if a.autocast else float("nan")
        t_o += time.time() - t0
        rows.append((i, lh, lo, la))
        print(f"step {i:3d}  HIP {lh:.5f}  oracle {lo:.5f}  diff {abs(lh - lo):.2e}" + (f"   autocast-policy oracle {la:.5f}  diff to fp32 {abs(la - lo):.2e}" if a.autocast else ""), flush=True)
    diffs = [abs(x - y) for _, x, y, _ in rows]
    if a.autocast:
        da = [abs(z - y) for _, _, y, z in rows]
        print(f"|autocast-policy oracle - fp32 oracle| (the reference's own precision policy): mean {sum(da) / len(da):.2e}, max {max(da):.2e}")
    num = den = 0.0
    for n in names:
        du = (model.arena.param(n).detach().cpu().double() - w[n].double()).reshape(-1)
        num += float((du ** 2).sum())
        den += float((w[n].double() ** 2).sum())
    print(f"steps {a.steps}: oracle loss {rows[0][2]:.4f} -> {rows[-1][2]:.4f}, HIP {rows[0][1]:.4f} -> {rows[-1][1]:.4f}; |HIP - oracle| mean {sum(diffs) / len(diffs):.2e}, "
          f"max {max(diffs):.2e}; parameters after the last step: rel L2 distance {(num / den) ** 0.5:.2e}; HIP {t_h:.1f}s, oracle {t_o:.1f}s")
    if a.out:
        with open(a.out, "w") as f:
            f.write("step,hip,oracle,oracle_autocast\n" + "".join(f"{i},{x:.6f},{y:.6f},{z:.6f}\n" for i, x, y, z in rows))


if __name__ == "__main__":
    main()
