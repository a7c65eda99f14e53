"""GPU parity of configs[1] at FULL WIDTH x FULL DEPTH against the oracle, once (VERDICT r3 "weak" #1 / item 2).

Every other comparison sees one of the two: the reference-made goldens run the real depth (32 / 6 / 32-36 layers) at tiny width,
tests/test_gpu_truewidth.py the real width at 4 / 2 / 2 layers, tests/test_gpu_fullsize.py the full model through size-independent
properties only.  Here the whole headline model — whisper-large-v3 (32 layers, d 1280, 1500 frames, taps 7/15/23/31), Q-Former 6L,
Llama-3.1-8B (32 layers, h 4096, inter 14336, 32 / 8 heads x 128, V 128 256, llama3 rope) — runs ONE batch (B = 1, S = 207 with
left padding) on the device and in `O.model_forward` + autograd (fp32) on the host cores with the same weights: loss, target-row
logits, audio features, the four tapped encoder states, decoder hidden states and EVERY connector gradient.

Yardstick for the gradients: per tensor, 2.5 x the error the reference's own autocast(bf16) policy carries on the tensor of the
same name at the same depth (tests/golden/autocast_policy_grad_errors.json["ref_deep_llama"], floor 5e-3), as in
tests/test_gpu_model.py::test_deep_and_tied_vs_reference_golden.  Reference: /root/reference/desta/models/modeling_desta25.py:758-938.
Cost on the GPU box: ~45 s of weight generation (8.8 B fp32 values on the host), ~40 s of oracle, ~35 GB of host memory."""
import json
import os
import time

import pytest
import torch

import desta_oracle as O
from helpers import cfg_from_dims, rel_err

pytestmark = pytest.mark.gpu


def _dims():
    return O.Dims(n_mels=128, enc_d=1280, enc_layers=32, enc_heads=20, enc_ffn=5120, enc_T=1500, taps=(7, 15, 23, 31),
                  qf_layers=6, qf_inter=3072, prompt_size=64, llm_h=4096, llm_layers=32, llm_hq=32, llm_hkv=8, llm_hd=128,
                  llm_inter=14336, vocab=128256, rms_eps=1e-5, rope_theta=500000.0, rope_llama3=(8.0, 1.0, 4.0, 8192), qk_norm=False,
                  tie_embeddings=False)


def test_full_width_full_depth_vs_oracle(golden_dir):
    from desta.models.modeling_desta25 import DeSTA25AudioModel
    torch.set_num_threads(max(1, min(64, len(os.sched_getaffinity(0)))))
    d = _dims()
    t0 = time.time()
    w = O.init_weights(d, seed=5)
    t_w = time.time() - t0
    batch = O.synthetic_batch(d, B=1, S_ctx=40, S_tgt=96, seed=9, pad=[7])
    t0 = time.time()
    model = DeSTA25AudioModel(cfg_from_dims(d), weights=w)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    names = O.trainable_names(d)

    # ---- product first (the device work is queued while the host starts the oracle): batch-major grid with the logits kept
    out = model(**batch, keep_logits=True)
    loss_full = out.loss.clone()
    logits = out.logits.float().cpu()
    af = model.connector.af.float().cpu().clone()
    taps = [model.enc_all[j].float().cpu().clone() for j in range(4)]
    hidden = [model.llm.xs[i + 1].float().cpu().clone() for i in (0, 15, 31)]
    model.backward()
    g_full = model.arena.grads.clone()
    # ... and the TRAINING FAST PATH (position-major grid, lm_head on target rows, fused rotary epilogue, dead-row skips)
    out2 = model(**batch)
    assert out2.logits is None
    model.arena.grads.zero_()                                                  # (the comparison below must not see the first backward's values)
    model.backward()
    g_fast = {n: model.arena.grad(n).detach().double().cpu() for n in names}

    # ---- oracle: fp32 forward + autograd on the host cores
    t0 = time.time()
    for n in names:
        w[n].requires_grad_(True)
    keep = {}
    loss_o, logits_o = O.model_forward(w, d, batch, keep)
    loss_o.backward()
    t_oracle = time.time() - t0
    grads_o = {n: w[n].grad.detach().double() for n in names}

    m = batch["attention_mask"].bool()
    tgt = torch.zeros_like(m)
    tgt[:, :-1] = batch["labels"][:, 1:] != -100                              # rows whose logits the loss reads (labels shifted left)
    rec = dict(loss=float(loss_o), dloss=abs(float(loss_full) - float(loss_o)), dloss_fast=abs(float(out2.loss) - float(loss_o)),
               logits=rel_err(logits[m], logits_o.detach()[m]), logits_target_rows=rel_err(logits[tgt], logits_o.detach()[tgt]),
               af=rel_err(af.view(1, d.prompt_size, d.llm_h), keep["audio_features"].detach()))
    rec["taps"] = [round(rel_err(taps[j].view(1, d.enc_T, d.enc_d), keep["taps"][j].detach()), 5) for j in range(4)]
    rec["hidden_1_16_32"] = [round(rel_err(h.view(1, -1, d.llm_h)[m], keep["llm_hidden"][i].detach()[m]), 5) for h, i in zip(hidden, (0, 15, 31))]
    a = torch.cat([g_fast[n].reshape(-1) for n in names])
    b = torch.cat([grads_o[n].reshape(-1) for n in names])
    rec["grad"], rec["cos"] = float((a - b).norm() / b.norm()), float((a @ b) / (a.norm() * b.norm()))
    rec["fast_vs_full_grad"] = float((model.arena.grads - g_full).double().norm() / g_full.double().norm())
    gn = sorted(float(grads_o[n].norm()) for n in names)
    floor = gn[len(gn) // 2] * 1e-2
    errs = {n: float((g_fast[n] - grads_o[n]).norm() / max(float(grads_o[n].norm()), floor)) for n in names}
    pol = json.load(open(os.path.join(golden_dir, "autocast_policy_grad_errors.json")))["ref_deep_llama"]
    ratio = {n: errs[n] / max(pol[n], 5e-3) for n in names}
    worst, wr = max(errs, key=errs.get), max(ratio, key=ratio.get)
    rec["worst_grad"] = (worst.split("connector.")[-1], round(errs[worst], 4))
    rec["worst_vs_policy"] = (wr.split("connector.")[-1], round(errs[wr], 4), round(pol[wr], 4))
    print("full width x full depth:", f"weights {t_w:.0f}s build {t_build:.0f}s oracle {t_oracle:.0f}s",
          {k: (round(v, 6) if isinstance(v, float) else v) for k, v in rec.items()})
    # bounds: the deep-golden ones (tests/test_gpu_model.py) — bf16 operands / fp32 accumulation against an fp32 oracle
    assert rec["dloss"] < 8e-3 and rec["dloss_fast"] < 8e-3, rec
    assert rec["logits"] < 4e-2 and rec["logits_target_rows"] < 4e-2 and rec["af"] < 2e-2, rec
    assert all(t < 1e-2 for t in rec["taps"]), rec                             # flat in depth: fp32 Whisper residual stream
    assert all(t < 3e-2 for t in rec["hidden_1_16_32"]), rec
    assert rec["grad"] < 5e-2 and rec["cos"] > 0.999, rec
    # the fast path changes WHICH rows are computed, not how, and the fused SwiGLU GEMMs take no split-K tails: bit-identical here
    # (measured 0.0); bounded at half of either path's own distance from the fp32 oracle
    assert rec["fast_vs_full_grad"] < 8e-3, rec
    assert ratio[wr] < 2.5, rec                                                # no tensor more than 2.5x the policy's own error
    assert errs[worst] < 0.06, rec
    # the same pair of passes with the SwiGLU epilogue fusion OFF (separate swiglu kernels, plain gate|up layout, K order of the
    # d(gate|up) GEMM un-permuted, split-K tails allowed): the same arithmetic in another fp32 summation order.  Over 32 decoder
    # layers of a random-init model such last-bit differences decorrelate the bf16 rounding noise of the two runs: measured 1.2e-2
    # between their gradients — each is 1.45e-2 from the fp32 oracle — and 7.7e-3 between this path's own fast / full grids (the last
    # layer's gate|up GEMM takes a split-K tail at 207 rows and none at the fast path's 97)
    model.llm.fuse_swiglu = False
    res = {}
    for fast in (False, True):
        o3 = model(**batch, keep_logits=True) if not fast else model(**batch)
        model.arena.grads.zero_()
        model.backward()
        res[fast] = (float(o3.loss), model.arena.grads.clone())
    model.llm.fuse_swiglu = True
    unf = dict(dloss_vs_fused=abs(res[False][0] - float(loss_full)), grad_vs_fused=float((res[False][1] - g_full).double().norm() / g_full.double().norm()),
               fast_vs_full=float((res[True][1] - res[False][1]).double().norm() / res[False][1].double().norm()))
    print("   SwiGLU fusion off:", unf)
    assert unf["dloss_vs_fused"] < 3e-3 and unf["grad_vs_fused"] < 2.5e-2 and unf["fast_vs_full"] < 1.5e-2, unf
