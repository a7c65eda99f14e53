"""GPU parity at TRUE WIDTH against the oracle: every other oracle / golden comparison runs at tiny width
(d = 128, h = 256, V = 512), so the accumulation lengths the headline config really has — K = 4096 / 14336 GEMM reductions,
D = 128 heads with a 4:1 GQA group, V = 128 256 / 151 936 cross-entropy rows, the [1280 x 3072] / [4096 x 1280] Adafactor
factors, 1500-frame cross-attention — were only covered by size-independent properties (tests/test_gpu_fullsize.py).

Here: whisper-large-v3 WIDTH with 4 encoder layers (all tapped), Q-Former 2L at d = 1280 / inter 3072 / 64 queries, and the
Llama-3.1-8B (h 4096, inter 14336, 32 / 8 heads, V 128 256, llama3 rope) resp. Qwen3-8B (inter 12288, V 151 936, q/k-norm)
WIDTH with 2 decoder layers; B = 1, S = 207 with left padding.  The oracle (fp32, autograd) runs the same weights and batch on
the host cores in well under a minute.  Tolerances are the deep-golden ones (tests/test_gpu_model.py): the product computes
in bf16 with fp32 accumulation like the reference under autocast, the oracle in fp32."""
import math
import time

import pytest
import torch

import desta_oracle as O
from helpers import cfg_from_dims, rel_err

pytestmark = pytest.mark.gpu

WIDTHS = {
    "llama31-8B": dict(llm_inter=14336, vocab=128256, rms_eps=1e-5, rope_theta=500000.0, rope_llama3=(8.0, 1.0, 4.0, 8192), qk_norm=False),
    "qwen3-8B": dict(llm_inter=12288, vocab=151936, rms_eps=1e-6, rope_theta=1e6, rope_llama3=None, qk_norm=True),
}


def _dims(which):
    return O.Dims(n_mels=128, enc_d=1280, enc_layers=4, enc_heads=20, enc_ffn=5120, enc_T=1500, taps=(0, 1, 2, 3),
                  qf_layers=2, qf_inter=3072, prompt_size=64, llm_h=4096, llm_layers=2, llm_hq=32, llm_hkv=8, llm_hd=128,
                  tie_embeddings=False, **WIDTHS[which])


@pytest.mark.parametrize("which", list(WIDTHS))
def test_true_width_shallow_depth_vs_oracle(which):
    from desta.models.modeling_desta25 import DeSTA25AudioModel
    from desta.optim import FusedAdafactor
    torch.set_num_threads(max(1, min(64, len(__import__("os").sched_getaffinity(0)))))
    d = _dims(which)
    t0 = time.time()
    w = O.init_weights(d, seed=11)
    batch = O.synthetic_batch(d, B=1, S_ctx=40, S_tgt=96, seed=3, pad=[7])
    model = DeSTA25AudioModel(cfg_from_dims(d), weights=w)
    t_build = time.time() - t0

    # ---- oracle: forward + autograd on the host cores (fp32)
    t0 = time.time()
    names = O.trainable_names(d)
    for n in names:
        w[n].requires_grad_(True)
    keep = {}
    loss_o, logits_o = O.model_forward(w, d, batch, keep)
    loss_o.backward()
    grads_o = {n: w[n].grad.detach().clone() for n in names}
    for n in names:
        w[n].requires_grad_(False)
        w[n].grad = None
    t_oracle = time.time() - t0

    # ---- product, batch-major path with the logits kept: activations / logits / loss
    out = model(**batch, keep_logits=True)
    m = batch["attention_mask"].bool()
    rec = dict(dloss=abs(float(out.loss) - float(loss_o)), logits=rel_err(out.logits.float().cpu()[m], logits_o.detach()[m]),
               af=rel_err(model.connector.af.float().view(1, d.prompt_size, d.llm_h), keep["audio_features"].detach()) if "audio_features" in keep
               else rel_err(model.llm.xs[0].float().view(1, -1, d.llm_h), keep["inputs_embeds"].detach()))
    rec["taps"] = [round(rel_err(model.enc_all[j].float().view(1, d.enc_T, d.enc_d), keep["taps"][j].detach()), 5) for j in range(4)]
    rec["hidden"] = [round(rel_err(model.llm.xs[i + 1].float().view(1, -1, d.llm_h)[m], hs.detach()[m]), 5) for i, hs in enumerate(keep["llm_hidden"])]
    model.backward()
    g_full = model.arena.grads.clone()

    # ---- product, TRAINING FAST PATH (position-major grid, lm_head on target rows, backward from the audio span)
    out2 = model(**batch)
    assert out2.logits is None
    rec["dloss_fast"] = abs(float(out2.loss) - float(loss_o))
    model.backward()
    a = torch.cat([model.arena.grad(n).reshape(-1).double().cpu() for n in names])
    b = torch.cat([grads_o[n].reshape(-1).double() for n in names])
    rec["grad"], rec["cos"] = float((a - b).norm() / b.norm()), float((a @ b) / (a.norm() * b.norm()))
    rec["fast_vs_full_grad"] = float((model.arena.grads - g_full).double().norm() / g_full.double().norm())
    gn = sorted(float(grads_o[n].double().norm()) for n in names)
    floor = gn[len(gn) // 2] * 1e-2
    errs = {n: float((model.arena.grad(n).double().cpu() - grads_o[n].double()).norm() / max(float(grads_o[n].double().norm()), floor)) for n in names}
    worst = max(errs, key=errs.get)
    rec["worst_grad"] = (worst.split("connector.")[-1], round(errs[worst], 4))
    print(which, f"build {t_build:.0f}s oracle {t_oracle:.0f}s", {k: (round(v, 6) if isinstance(v, float) else v) for k, v in rec.items()})
    assert rec["dloss"] < 8e-3 and rec["dloss_fast"] < 8e-3, rec
    assert rec["logits"] < 4e-2 and rec["af"] < 2e-2, rec
    assert all(t < 1e-2 for t in rec["taps"]) and all(t < 2e-2 for t in rec["hidden"]), rec
    assert rec["grad"] < 5e-2 and rec["cos"] > 0.999, rec
    assert rec["fast_vs_full_grad"] < 5e-3, rec
    assert errs[worst] < 0.15, rec

    # ---- one clip + Adafactor step on the REAL factor shapes ([1280 x 3072], [3072 x 1280], [4096 x 1280], [1, 64, 1280] prompts)
    # against the oracle's restatement of transformers.Adafactor fed with the SAME (device-produced) gradients
    opt = FusedAdafactor(model.arena, weight_decay=0.01, max_grad_norm=1.0)
    p0 = {n: model.arena.param(n).detach().cpu().clone() for n in names}
    g_host = [model.arena.grad(n).detach().cpu().clone() for n in names]
    lr = 1e-3
    opt.step(lr)
    torch.cuda.synchronize()
    params = [p0[n].clone() for n in names]
    st = O.adafactor_init(params)
    O.clip_grad_norm(g_host, 1.0, f64=True)
    wd = [0.01 if dm else 0.0 for dm in O.decay_mask(names)]
    O.adafactor_step(params, g_host, st, lr, wd, f64_stats=True)
    worst_u = 0.0
    for n, p_ref in zip(names, params):
        got = model.arena.param(n).detach().cpu()
        upd_ref = (p_ref - p0[n]).double()
        e = float((got.double() - p_ref.double()).norm() / max(float(upd_ref.norm()), 1e-30))
        worst_u = max(worst_u, e)
        assert e < 2e-3, (n, e)                            # error of the UPDATE itself (fp32 row / column statistics over up to 3.9 M elements)
    print(which, "adafactor worst update error", worst_u)


def test_true_width_lora_adapters_vs_oracle():
    """`use_lora=True` at the Llama-3.1-8B width (h 4096, q/k/v 4096 / 1024 / 1024, r = 16; 2 decoder layers): loss and every
    gradient — adapters and connector — against fp32 autograd on the host (parity unpinned against peft itself, which is absent:
    see tests/test_gpu_lora.py).  S = 207 is not a multiple of 64: the adapter gradients' token reductions run on zero-padded rows."""
    from desta.models.modeling_desta25 import DeSTA25AudioModel
    torch.set_num_threads(max(1, min(64, len(__import__("os").sched_getaffinity(0)))))
    d = _dims("llama31-8B")
    d.lora_r = 16
    w = O.init_weights(d, seed=11)
    batch = O.synthetic_batch(d, B=1, S_ctx=40, S_tgt=96, seed=3, pad=[7])
    model = DeSTA25AudioModel(cfg_from_dims(d, use_lora=True, lora_dropout=0.0), weights=w)
    names = O.trainable_names(d)
    assert set(names) == set(model.trainable_parameter_names)
    for n in names:
        w[n].requires_grad_(True)
    loss_o, _ = O.model_forward(w, d, batch)
    loss_o.backward()
    grads_o = {n: w[n].grad.detach().clone() for n in names}
    out = model(**batch)
    model.backward()
    a = torch.cat([model.arena.grad(n).reshape(-1).double().cpu() for n in names])
    b = torch.cat([grads_o[n].reshape(-1).double() for n in names])
    lora = [n for n in names if ".lora_" in n]
    al = torch.cat([model.arena.grad(n).reshape(-1).double().cpu() for n in lora])
    bl = torch.cat([grads_o[n].reshape(-1).double() for n in lora])
    errs = {n: float((model.arena.grad(n).double().cpu() - grads_o[n].double()).norm() / float(grads_o[n].double().norm())) for n in lora}
    worst = max(errs, key=errs.get)
    rec = dict(dloss=abs(float(out.loss) - float(loss_o)), grad=float((a - b).norm() / b.norm()), cos=float((a @ b) / (a.norm() * b.norm())),
               lora_grad=float((al - bl).norm() / bl.norm()), worst_lora=(worst, round(errs[worst], 4)))
    print("lora true width", rec)
    assert rec["dloss"] < 8e-3 and rec["grad"] < 5e-2 and rec["cos"] > 0.999 and rec["lora_grad"] < 5e-2 and errs[worst] < 0.1, rec
