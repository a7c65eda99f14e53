"""GPU parity of the whole hot path (HIP kernels behind the C ABI, driven by the desta package) against
  (1) the golden vectors produced by the REFERENCE's own classes (tests/golden/ref_tiny_*.safetensors), and
  (2) the CPU oracle (oracle/desta_oracle.py) on seeded inputs: per-stage activations, logits, loss,
      connector gradients, and parameters after several optimizer steps.
Tolerances: the product computes in bf16 (fp32 accumulate / statistics) like the reference under
autocast; the oracle / goldens are fp32.  Stated per check below."""
import copy
import os
import math

import pytest
import torch

import desta_oracle as O
from helpers import cfg_from_dims, golden_batch, rel_err

pytestmark = pytest.mark.gpu


def _model(d, seed=7):
    from desta.models.modeling_desta25 import DeSTA25AudioModel
    w = O.init_weights(d, seed=seed)
    return DeSTA25AudioModel(cfg_from_dims(d), weights=w), w


@pytest.mark.parametrize("name", ["llama", "qwen3"])
def test_forward_backward_vs_reference_golden(golden_dir, name):
    d = O.tiny_dims(name == "qwen3")
    g, batch = golden_batch(golden_dir, name)
    model, w = _model(d)
    out = model(**batch, keep_logits=True)
    loss = float(out.loss)
    # bf16 path vs the reference's fp32 golden; bounds = 3x what the path measures (round 3: dloss 6e-4, features 3e-3, logits 6e-3;
    # rounds 1-3 asserted 2e-2 / 2e-2 / 3e-2, which a tenfold regression would have passed)
    assert abs(loss - float(g["loss"])) < 2e-3, (loss, float(g["loss"]))
    m = g["attention_mask"].bool()
    e_logits = rel_err(out.logits.float().cpu()[m], g["logits"][m])
    e_af = rel_err(model.connector.af.float().view(g["audio_features"].shape), g["audio_features"])
    assert e_af < 1e-2, e_af
    assert e_logits < 2e-2, e_logits
    model.backward()
    errs = {}
    # relative L2 error per tensor; tensors whose true gradient is (numerically) zero — the key biases:
    # softmax is invariant to a per-query constant — are measured against the typical gradient norm
    gn = sorted(float(g["grad::" + n].double().norm()) for n in model.trainable_parameter_names)
    floor = gn[len(gn) // 2] * 1e-2
    for n in model.trainable_parameter_names:
        ref = g["grad::" + n].double()
        got = model.arena.grad(n).double().cpu()
        errs[n] = float((got - ref).norm() / max(float(ref.norm()), floor))
    for n in sorted(errs, key=errs.get)[-6:]:
        print(f"   grad err {errs[n]:.4f}  {n}")
    worst = max(errs, key=errs.get)
    # per-TENSOR yardstick, as in the deep test below: the error the reference's own autocast(bf16) policy carries on that tensor
    # against the same fp32 golden (tests/golden/autocast_policy_grad_errors.json); no tensor more than 2.5x that (floor 5e-3)
    import json
    pol = json.load(open(os.path.join(golden_dir, "autocast_policy_grad_errors.json")))["ref_tiny_" + name]
    ratio = {n: errs[n] / max(pol[n], 5e-3) for n in errs}
    wr = max(ratio, key=ratio.get)
    print("loss", loss, float(g["loss"]), "logits", e_logits, "af", e_af, "worst grad", worst, errs[worst],
          "worst vs policy", wr, errs[wr], pol[wr])
    assert ratio[wr] < 2.5, (wr, errs[wr], pol[wr])
    assert errs[worst] < 4e-2, (worst, errs[worst])
    # global gradient direction: cosine over the whole arena
    a = torch.cat([model.arena.grad(n).reshape(-1).double().cpu() for n in model.trainable_parameter_names])
    b = torch.cat([g["grad::" + n].reshape(-1).double() for n in model.trainable_parameter_names])
    cos = float((a @ b) / (a.norm() * b.norm()))
    assert cos > 0.999, cos


DEEP = {"ref_deep_llama": lambda: O.deep_dims(False), "ref_deep_qwen3": lambda: O.deep_dims(True), "ref_tied_qwen3": O.tied_dims}


@pytest.mark.parametrize("name", list(DEEP))
def test_deep_and_tied_vs_reference_golden(golden_dir, name):
    """Parity at the reference's REAL depth (32 encoder layers tapped at 7/15/23/31, Q-Former 6L, 32 / 36 decoder layers) and
    on the Qwen3-4B-like geometry (tied lm_head, Hq*hd != hidden), against goldens made by the reference's own forward +
    autograd in fp32.  Yardstick: the reference's own autocast(bf16) policy restated on the CPU differs from the same goldens
    by dloss 1.0e-3 / 2.2e-3, logits 1.5e-2, audio features 6e-3, taps 5.7e-3, whole-arena gradient 1.5e-2 / 1.9e-2
    (tests/test_oracle_pin.py::test_autocast_policy_error_vs_depth); the bounds below are 2-3x that."""
    d = DEEP[name]()
    g, batch = golden_batch(golden_dir, name)
    model, w = _model(d)
    out = model(**batch, keep_logits=True)
    loss = float(out.loss)
    m = g["attention_mask"].bool()
    rec = dict(dloss=abs(loss - float(g["loss"])), logits=rel_err(out.logits.float().cpu()[m], g["logits"][m]),
               af=rel_err(model.connector.af.float().view(g["audio_features"].shape), g["audio_features"]))
    if "tap_states" in g:
        B, T = g["tap_states"].shape[1:3]
        rec["taps"] = [round(rel_err(model.enc_all[j].float().view(B, T, d.enc_d), g["tap_states"][j]), 5) for j in range(len(d.taps))]
    model.backward()
    names = model.trainable_parameter_names
    a = torch.cat([model.arena.grad(n).reshape(-1).double().cpu() for n in names])
    b = torch.cat([g["grad::" + n].reshape(-1).double() for n in names])
    rec["grad"], rec["cos"] = float((a - b).norm() / b.norm()), float((a @ b) / (a.norm() * b.norm()))
    gn = sorted(float(g["grad::" + n].double().norm()) for n in names)
    floor = gn[len(gn) // 2] * 1e-2
    errs = {n: float((model.arena.grad(n).double().cpu() - g["grad::" + n].double()).norm() / max(float(g["grad::" + n].double().norm()), floor)) for n in names}
    worst = max(errs, key=errs.get)
    rec["worst_grad"] = (worst.split("connector.")[-1], round(errs[worst], 4))
    # per-TENSOR yardstick: the error the reference's own autocast(bf16) policy carries on that tensor against the same fp32
    # golden (tests/golden/autocast_policy_grad_errors.json, made by tests/golden/make_policy_grad_errors.py on the CPU).  Round 2
    # showed 10-13 % on the deep cross-attention query weights against 2.5 % for the policy: delta = rowsum(dO * O) taken from
    # the bf16-ROUNDED attention output; with the fp32 output (desta_attn_desc.O_f32) the tensor sits at the policy's floor.
    import json
    pol = json.load(open(os.path.join(golden_dir, "autocast_policy_grad_errors.json")))[name]
    ratio = {n: errs[n] / max(pol[n], 5e-3) for n in names}
    wr = max(ratio, key=ratio.get)
    rec["worst_vs_policy"] = (wr.split("connector.")[-1], round(errs[wr], 4), round(pol[wr], 4))
    print(name, {k: (round(v, 5) if isinstance(v, float) else v) for k, v in rec.items()})
    assert ratio[wr] < 2.5, rec                                              # no tensor more than 2.5x the policy's own error (floor 5e-3)
    assert rec["dloss"] < 8e-3, rec
    assert rec["logits"] < 4e-2 and rec["af"] < 2e-2, rec
    assert all(t < 1e-2 for t in rec.get("taps", [])), rec                 # flat in depth: fp32 Whisper residual stream
    assert rec["grad"] < 5e-2 and rec["cos"] > 0.999, rec
    assert errs[worst] < 0.06, rec                                           # (round 2: 0.15, with the 10-13 % outlier inside)


@pytest.mark.parametrize("s_tgt", [150, 80])
def test_rope_fused_into_the_qkv_epilogue(s_tgt):
    """Sequences of >= 128 tokens take the rotary embedding inside the q|k|v GEMM epilogue (permuted frozen weights, adjacent
    pairs) and its transpose inside the attention backward's dQ / dK stores; shorter ones the separate kernel.  Both must agree
    with each other (same loss to bf16 rounding, same gradients) and with the oracle, on both token layouts."""
    d = O.tiny_dims(False)
    model, w = _model(d)
    batch = O.synthetic_batch(d, B=3, S_ctx=9, S_tgt=s_tgt, seed=5, pad=[0, 7, 2])
    S = batch["input_ids"].shape[1]
    names = model.trainable_parameter_names
    loss_o, _ = O.model_forward(w, d, batch)
    res = {}
    for fuse in (True, False):
        model.llm.fuse_rope = fuse
        for fast in (True, False):                                  # position-major training fast path / batch-major grid
            out = model(**batch) if fast else model(**batch, keep_logits=True)
            assert model.llm._rope_fused == (fuse and S >= 128)
            model.backward()
            res[(fuse, fast)] = (float(out.loss), model.arena.grads.clone())
            assert abs(float(out.loss) - float(loss_o)) < 2e-2
    for fast in (True, False):
        la, ga = res[(True, fast)]
        lb, gb = res[(False, fast)]
        assert abs(la - lb) < 5e-3, (la, lb)
        assert float((ga - gb).double().norm() / gb.double().norm()) < 2e-2
    model.llm.fuse_rope = True


def test_stagewise_vs_oracle():
    d = O.tiny_dims(False)
    model, w = _model(d)
    batch = O.synthetic_batch(d, B=3, S_ctx=9, S_tgt=40, seed=5, pad=[0, 7, 2])
    keep = {}
    loss_o, logits_o = O.model_forward(w, d, batch, keep)
    out = model(**batch, keep_logits=True)
    T = d.enc_T
    for j, tap in enumerate(keep["taps"]):
        e = rel_err(model.enc_all[j].float().view(3, T, d.enc_d), tap)
        assert e < 2e-2, (j, e)
    e = rel_err(model.connector.qf_out.view(len(d.taps), 3, d.prompt_size, d.enc_d), torch.stack(keep["qformer_out"]))
    assert e < 2e-2, e
    e = rel_err(model.llm.xs[0].float().view(3, -1, d.llm_h), keep["inputs_embeds"])
    assert e < 1e-2, e
    m = batch["attention_mask"].bool()
    for i, hs in enumerate(keep["llm_hidden"]):
        e = rel_err(model.llm.xs[i + 1].float().view(3, -1, d.llm_h)[m], hs[m])
        assert e < 2e-2, (i, e)
    assert abs(float(out.loss) - float(loss_o)) < 2e-2
    assert rel_err(out.logits.float().cpu()[m], logits_o[m]) < 3e-2


def test_no_audio_and_eval_paths():
    d = O.tiny_dims(False)
    model, w = _model(d)
    batch = O.synthetic_batch(d, B=2, S_ctx=6, S_tgt=10, seed=2)
    plain = {"input_ids": batch["input_ids"], "attention_mask": batch["attention_mask"], "labels": batch["labels"],
             "batch_features": None, "batch_transcription_ids": [], "batch_start_positions": []}
    out = model(**plain)
    x = O.embed_splice(w, d, batch["input_ids"], None, [], [])
    lo = O.causal_lm_loss(O.llm_forward(w, d, x, batch["attention_mask"]), batch["labels"])
    assert abs(float(out.loss) - float(lo)) < 2e-2
    with pytest.raises(RuntimeError):
        model.backward()
    model.eval()
    out = model(**batch)                      # eval: logits stay logits
    _, logits_o = O.model_forward(w, d, batch)
    assert rel_err(out.logits.float().cpu(), logits_o) < 3e-2
    bad = dict(batch)
    bad["batch_features"] = batch["batch_features"][..., :-2]
    with pytest.raises(ValueError, match="Whisper expects the mel input features"):
        model(**bad)


def test_transcription_splice():
    """Non-empty transcription ids are spliced after the 64 audio rows (modeling_desta25.py:1034-1041)."""
    d = O.tiny_dims(False)
    model, w = _model(d)
    model.eval()
    batch = O.synthetic_batch(d, B=2, S_ctx=4, S_tgt=20, seed=3)
    batch["batch_transcription_ids"] = [torch.tensor([[5, 6, 7]]), torch.zeros(1, 0, dtype=torch.long)]
    keep = {}
    O.model_forward(w, d, batch, keep)
    model(**batch)
    assert rel_err(model.llm.xs[0].float().view(2, -1, d.llm_h), keep["inputs_embeds"]) < 1e-2


@pytest.mark.parametrize("name", ["llama", "qwen3"])
def test_train_steps_vs_oracle(name):
    """3 optimizer steps (forward, backward, clip 1.0, Adafactor, linear warmup) vs the oracle."""
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments
    d = O.tiny_dims(name == "qwen3")
    model, w = _model(d, seed=11)
    w = {k: v.clone() for k, v in w.items()}
    args = TrainingArguments(learning_rate=1e-3, warmup_steps=2, max_steps=10, logging_steps=1, overlap_comm=(name == "llama"))
    tr = DeSTA25Trainer(model, args=args)
    names = O.trainable_names(d)
    st = O.adafactor_init([w[n] for n in names])
    batches = [O.synthetic_batch(d, B=2, S_ctx=5, S_tgt=24, seed=100 + i, pad=[0, i]) for i in range(3)]
    losses = tr.train(batches)
    ref_losses = []
    for i, b in enumerate(batches):
        lr = O.linear_warmup_lr(i, 1e-3, 2, 10)
        lo, _, _, _ = O.train_step(w, d, b, st, lr)
        ref_losses.append(float(lo))
        assert abs(losses[i] - float(lo)) < 2e-2, (i, losses[i], float(lo))
    assert len(set(losses)) == len(losses), losses            # per-step values, not aliases of one buffer
    # step-to-step differences follow the oracle (tighter than the absolute bf16-vs-fp32 offset)
    for i in range(1, len(losses)):
        assert abs((losses[i] - losses[0]) - (ref_losses[i] - ref_losses[0])) < 1e-2
    sd = model.state_dict()
    # parameters moved by ~lr per step; compare the UPDATE (p_after - p_before) direction and size
    w0 = O.init_weights(d, seed=11)
    num = den = 0.0
    for n in names:
        du = (sd[n].cpu().double() - w0[n].double()).reshape(-1)
        do = (w[n].double() - w0[n].double()).reshape(-1)
        num += float(((du - do) ** 2).sum())
        den += float((do ** 2).sum())
    rel = (num / den) ** 0.5
    print("update rel err", rel)
    assert rel < 0.15, rel
    assert tr.log_history and "train/lm_loss" in tr.log_history[0] and "train/ppl" in tr.log_history[0]


def test_checkpoint_roundtrip(tmp_path):
    from desta.models.modeling_desta25 import DeSTA25AudioModel
    from safetensors.torch import load_file
    d = O.tiny_dims(False)
    model, w = _model(d)
    model.save_pretrained(str(tmp_path))
    sd = load_file(str(tmp_path / "model.safetensors"))
    assert sorted(sd.keys()) == sorted(O.trainable_names(d))                    # trainable-only, reference key names
    for n in O.trainable_names(d):
        assert torch.equal(sd[n], w[n])
    m2 = DeSTA25AudioModel.from_pretrained(str(tmp_path), weights={k: v for k, v in w.items() if "connector" not in k})
    for n in O.trainable_names(d):
        assert torch.equal(m2.arena.param(n).cpu(), w[n])


def test_qformer_dropout_training_semantics():
    """Dropout 0.1 (BertConfig default, hazard H3) is live in train mode only, reproducible per forward count,
    and the hand-written backward differentiates THROUGH the same masks (directional finite difference)."""
    from desta.models.modeling_desta25 import DeSTA25AudioModel
    d = O.tiny_dims(False)
    w = O.init_weights(d, seed=7)
    model = DeSTA25AudioModel(cfg_from_dims(d, dropout=0.1), weights=w)
    ref = DeSTA25AudioModel(cfg_from_dims(d, dropout=0.0), weights=w)
    batch = O.synthetic_batch(d, B=2, S_ctx=5, S_tgt=24, seed=8)
    model.eval(); ref.eval()
    assert float(model(**batch).loss) == float(ref(**batch).loss)             # eval: no dropout
    model.train(); ref.train()
    l_ref = float(ref(**batch).loss)

    def loss_at(count):
        model._fwd_count = count
        model.mark_weights_updated()
        return model(**batch)
    l1, l1b, l2 = float(loss_at(5).loss), float(loss_at(5).loss), float(loss_at(6).loss)
    assert l1 == l1b and l1 != l2 and l1 != l_ref                             # same masks <-> same forward count
    # directional derivative along the gradient with the masks of forward #5 held fixed
    loss_at(5)
    model.backward()
    gvec = model.arena.grads.clone()
    gn = float(gvec.norm())
    assert math.isfinite(gn) and gn > 0
    p0 = model.arena.params.clone()
    eps = 2e-2 / gn
    model.arena.params.copy_(p0 + eps * gvec)
    lp = float(loss_at(5).loss)
    model.arena.params.copy_(p0 - eps * gvec)
    lm = float(loss_at(5).loss)
    model.arena.params.copy_(p0)
    fd, an = (lp - lm) / (2 * eps), gn * gn
    print("directional derivative: finite-diff", fd, "analytic", an)
    assert abs(fd - an) < 0.2 * an + 2e-3 / eps * 0.5                         # bf16 loss noise ~1e-3 over a 2*eps*|g|^2 = 4e-2 signal


def test_hf_checkpoint_wire_format_and_resume(tmp_path):
    """checkpoint-<step>/ written by the HIP trainer: (1) optimizer.pt loads into a REAL transformers Adafactor
    built the way HF Trainer builds it (decay / no-decay groups over named_parameters order) and both continue
    identically; (2) resuming the HIP trainer from the directory reproduces an uninterrupted run bit for bit."""
    import json
    from safetensors.torch import load_file
    from transformers.optimization import Adafactor
    from desta.models.modeling_desta25 import DeSTA25AudioModel, reference_parameter_names
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments
    d = O.tiny_dims(False)
    w = O.init_weights(d, seed=5)
    batches = [O.synthetic_batch(d, B=2, S_ctx=5, S_tgt=24, seed=300 + i) for i in range(4)]
    args = TrainingArguments(learning_rate=1e-3, warmup_steps=2, max_steps=10, logging_steps=1, overlap_comm=False)

    def fresh(dropout):
        m = DeSTA25AudioModel(cfg_from_dims(d, dropout=dropout), weights=w)
        return m, DeSTA25Trainer(m, args=args)
    # uninterrupted 4 steps (dropout ON: the stream position must survive the checkpoint)
    m_a, t_a = fresh(0.1)
    la = t_a.train(batches)
    # 2 steps, checkpoint, resume in a NEW model/trainer, 2 more steps
    m_b, t_b = fresh(0.1)
    lb = t_b.train(batches[:2])
    ck = str(tmp_path / "checkpoint-2")
    t_b.save_checkpoint(ck)
    assert sorted(os.listdir(ck)) == ["config.json", "desta_hip_state.json", "model.safetensors", "optimizer.pt", "rng_state.pth", "scheduler.pt",
                                      "trainer_state.json", "training_args.bin"]
    from transformers.trainer_callback import TrainerState
    assert TrainerState.load_from_json(os.path.join(ck, "trainer_state.json")).global_step == 2       # HF reads what this trainer wrote
    m_c, t_c = fresh(0.1)
    t_c.resume_from_checkpoint(ck)
    assert t_c.global_step == 2 and json.load(open(os.path.join(ck, "trainer_state.json")))["global_step"] == 2
    lc = t_c.train(batches[2:])
    assert lb + lc == la
    assert torch.equal(m_c.arena.params, m_a.arena.params)
    # HF interop: feed optimizer.pt to transformers' Adafactor and take one identical step on both sides
    names = reference_parameter_names(m_b.config)
    sdm = load_file(os.path.join(ck, "model.safetensors"))
    params = {n: torch.nn.Parameter(sdm[n].clone()) for n in names}
    dm = dict(zip(names, O.decay_mask(names)))
    hf = Adafactor([{"params": [params[n] for n in names if dm[n]], "weight_decay": 0.01},
                    {"params": [params[n] for n in names if not dm[n]], "weight_decay": 0.0}],
                   lr=1e-3, scale_parameter=False, relative_step=False)
    hf.load_state_dict(torch.load(os.path.join(ck, "optimizer.pt"), weights_only=True))
    m_d, t_d = fresh(0.0)
    t_d.resume_from_checkpoint(ck)
    g = torch.Generator().manual_seed(9)
    for n in names:
        gr = 0.01 * torch.randn(params[n].shape, generator=g)
        params[n].grad = gr.clone()
        m_d.arena.grad(n).copy_(gr)
    torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
    lr = O.linear_warmup_lr(2, 1e-3, 2, 10)
    for gq in hf.param_groups:
        gq["lr"] = lr
    hf.step()
    t_d.optimizer.step(lr)
    for n in names:
        torch.testing.assert_close(m_d.arena.param(n).cpu(), params[n].detach(), rtol=1e-5, atol=1e-6)


def test_connector_standalone_vs_reference_golden(golden_dir):
    """`connector(list_of_encoder_states)` like the reference's own unit test (tests/test_modeling.py:21-36), but
    with numbers: the golden holds the reference QformerConnector's output on seeded random states."""
    d = O.tiny_dims(False)
    g, _ = golden_batch(golden_dir, "llama")
    model, w = _model(d)
    states = list(g["conn_states"].unbind(0))
    out = model.connector(states)
    assert out.shape == g["conn_out"].shape == (2, d.prompt_size, d.llm_h)
    assert rel_err(out.float(), g["conn_out"]) < 2e-2


def test_connector_reference_unit_test_shapes_fwd_bwd():
    """The shapes of the reference's connector unit test (whisper-tiny width 384 / 6 heads, prompt_size 4, 100 frames,
    2 Q-Former layers, BertConfig-default intermediate 3072): ragged everything (Sq = 4, Sk = 100, R = 32 rows).
    Forward and all gradients against the oracle's autograd."""
    import dataclasses
    d = dataclasses.replace(O.tiny_dims(False), enc_d=384, enc_heads=6, enc_ffn=1536, enc_T=100, qf_inter=3072, prompt_size=4)
    model, w = _model(d, seed=3)
    gen = torch.Generator().manual_seed(5)
    B = 2
    states = [torch.randn(B, d.enc_T, d.enc_d, generator=gen).to(torch.bfloat16).float() for _ in range(d.enc_layers)]
    names = [n for n in O.trainable_names(d)]
    for n in names:
        w[n].requires_grad_(True)
    ref = O.mix_proj(w, d, [O.qformer(w, d, j, states[t]) for j, t in enumerate(d.taps)])
    out = model.connector(states)
    assert out.shape == (B, 4, d.llm_h)
    assert rel_err(out.float(), ref) < 2e-2
    d_af = torch.randn(B * 4, d.llm_h, generator=gen).to(torch.bfloat16)
    ref.backward(d_af.float().view(B, 4, d.llm_h))
    model.connector.backward(d_af.cuda())
    gn = sorted(float(w[n].grad.double().norm()) for n in names)
    floor = gn[len(gn) // 2] * 1e-2
    worst = 0.0
    for n in names:
        r = w[n].grad.double()
        e = float((model.arena.grad(n).double().cpu() - r).norm() / max(float(r.norm()), floor))
        worst = max(worst, e)
        assert e < 8e-2, (n, e)
    print("worst connector grad err", worst)


def test_loss_curve_tracks_oracle_over_many_steps():
    """120 optimizer steps on a cycled pool of 8 batches (tiny config, dropout off): the bf16 HIP path's loss curve
    stays on the fp32 oracle's (north star: loss curve parity over many steps).  Bounds = the measured regime of the
    1000-step three-way runs of tools/loss_curve.py (profiles/r02_loss_curve_summary.log): per-step mean |d| of the HIP path
    against the fp32 oracle 1.2e-3 (Llama) / 1.7e-3 (Qwen3), the reference's OWN autocast(bf16) policy against the same fp32
    run 2.2e-3 / 1.5e-3 — the bf16 floor; single steps reach 1.5e-2; the smoothed curves agree within 4e-4."""
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments
    torch.set_num_threads(min(8, torch.get_num_threads()))
    d = O.tiny_dims(False)
    n_steps, lr, warm = 120, 1e-3, 10
    model, w = _model(d, seed=21)
    w = {k: v.clone() for k, v in w.items()}
    tr = DeSTA25Trainer(model, args=TrainingArguments(learning_rate=lr, warmup_steps=warm, max_steps=n_steps, logging_steps=10 ** 9))
    names = O.trainable_names(d)
    st = O.adafactor_init([w[n] for n in names])
    pool = [O.synthetic_batch(d, B=2, S_ctx=5, S_tgt=24, seed=500 + i, pad=[0, i % 3]) for i in range(8)]
    hip = tr.train([pool[i % 8] for i in range(n_steps)])
    # THREE curves on the same batches: fp32 oracle, the oracle under the reference's own autocast(bf16) policy (what the reference
    # itself trains under, hazard H11), HIP
    wp = {k: v.clone() for k, v in w.items()}
    stp = O.adafactor_init([wp[n] for n in names])
    ref, polc = [], []
    for i in range(n_steps):
        lr_i = O.linear_warmup_lr(i, lr, warm, n_steps)
        lo, _, _, _ = O.train_step(w, d, pool[i % 8], st, lr_i)
        lp, _, _, _ = O.train_step(wp, d, pool[i % 8], stp, lr_i, autocast=True)
        ref.append(float(lo))
        polc.append(float(lp))
    mean = lambda v: sum(v) / len(v)
    diff = [abs(a - b) for a, b in zip(hip, ref)]
    dpol = [abs(a - b) for a, b in zip(hip, polc)]
    dpr = [abs(a - b) for a, b in zip(polc, ref)]
    print("loss", ref[0], "->", ref[-1], "| HIP - fp32: mean", mean(diff), "max", max(diff), "| HIP - autocast policy: mean", mean(dpol), "max", max(dpol),
          "| policy - fp32: mean", mean(dpr), "max", max(dpr))
    assert ref[-1] < ref[0] - 0.3                                # the pool is being fitted
    # round 3 measured mean 3.7e-4 / max 1.6e-3 against fp32 (with the fp32 attention output feeding delta); 3x that
    assert mean(diff) < 1.2e-3 and max(diff) < 5e-3, (mean(diff), max(diff))
    # against the policy oracle — two bf16 runs with different rounding points — and relative to the policy's own distance from fp32:
    # the HIP path is no farther from fp32 than 1.5x the reference's own precision policy is
    assert mean(dpol) < 2.5e-3 and max(dpol) < 2e-2, (mean(dpol), max(dpol))
    assert mean(diff) < 1.5 * mean(dpr) + 2e-4, (mean(diff), mean(dpr))
    assert abs(sum(hip[-10:]) - sum(ref[-10:])) / 10 < 1e-3


def test_training_entry_point_debug_config(tmp_path):
    """The reference's integration test is its run recipe (`train_desta.py --config-name desta25_debug +dataset=debug`):
    same command line here, synthetic data, 12 steps; checks the experiment directory layout the reference produces."""
    import importlib.util
    import json
    import os
    from safetensors.torch import load_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_desta", os.path.join(root, "examples", "train", "train_desta.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    exp = str(tmp_path / "exp")
    m.main(["--config-name", "desta25_debug", "+dataset=debug", f"exp_dir={exp}", "trainer.max_steps=12"])
    assert os.path.isfile(os.path.join(exp, "config.yaml"))
    init = load_file(os.path.join(exp, "checkpoint-initial", "model.safetensors"))
    last = load_file(os.path.join(exp, "checkpoint-12", "model.safetensors"))
    assert sorted(init) == sorted(last) and all(k.startswith("perception.connector.") for k in init)   # trainable-only, reference keys
    assert any(not torch.equal(init[k], last[k]) for k in init)                                          # parameters moved
    cfgj = json.load(open(os.path.join(exp, "checkpoint-12", "config.json")))
    assert cfgj["model_type"] == "desta25" and cfgj["connector_mode"] == "qformer_1"
    assert os.path.isfile(os.path.join(exp, "checkpoint-12", "optimizer.pt"))


@pytest.mark.parametrize("name", ["llama", "qwen3"])
def test_compact_lm_head_equals_full_grid(golden_dir, name):
    """Training forward / backward with lm_head + CE restricted to the target rows == the full-grid path: same loss, same
    connector gradients (only the fp32 summation order of split-K tail tiles may differ)."""
    d = O.tiny_dims(name == "qwen3")
    g, batch = golden_batch(golden_dir, name)
    model, w = _model(d)
    res = {}
    for compact in (True, False):
        model.compact_lm_head = compact
        model.mark_weights_updated()
        out = model(**batch)
        assert out.logits is None
        assert (model.llm.compact is not None) == compact and model.llm.s_major == compact
        model.backward()
        res[compact] = (float(out.loss), model.arena.grads.clone())
    assert abs(res[True][0] - res[False][0]) < 1e-6 * max(1.0, abs(res[False][0]))
    a, b = res[True][1].double(), res[False][1].double()
    assert float((a - b).norm() / b.norm()) < 2e-3
    assert abs(res[True][0] - float(g["loss"])) < 2e-2
    # keep_logits / eval still take the full (batch-major) grid
    model.compact_lm_head = True
    out = model(**batch, keep_logits=True)
    assert out.logits is not None and model.llm.compact is None and not model.llm.s_major
    # a batch without any target: loss 0, all gradients 0 (ForCausalLMLoss of an all-ignored batch)
    nb = dict(batch)
    nb["labels"] = torch.full_like(batch["labels"], -100)
    out = model(**nb)
    assert float(out.loss) == 0.0
    model.backward()
    assert float(model.arena.grads.abs().max()) == 0.0


@pytest.mark.parametrize("name", ["llama", "qwen3"])
def test_training_fast_path_vs_reference_golden(golden_dir, name):
    """The training fast path (position-major grid, lm_head on target rows, backward from the first audio span) pinned
    directly to the reference's own loss and gradients (left-padded rows, different audio starts per sample)."""
    d = O.tiny_dims(name == "qwen3")
    g, batch = golden_batch(golden_dir, name)
    model, w = _model(d)
    out = model(**batch)
    assert out.logits is None and model.llm.s_major and model.llm.compact is not None
    assert abs(float(out.loss) - float(g["loss"])) < 2e-2
    model.backward()
    names = model.trainable_parameter_names
    gn = sorted(float(g["grad::" + n].double().norm()) for n in names)
    floor = gn[len(gn) // 2] * 1e-2
    worst = max(float((model.arena.grad(n).double().cpu() - g["grad::" + n].double()).norm() / max(float(g["grad::" + n].double().norm()), floor))
                for n in names)
    assert worst < 8e-2, worst
    a = torch.cat([model.arena.grad(n).reshape(-1).double().cpu() for n in names])
    b = torch.cat([g["grad::" + n].reshape(-1).double() for n in names])
    assert float((a @ b) / (a.norm() * b.norm())) > 0.999


def test_prefetch_hit_needs_the_same_tensor_object_and_training_empty_batch():
    """(a) ADVICE r1: a prefetched encoder output is reused only for the very tensor object that was prefetched — a NEW tensor of the
    same shape (possibly at the freed tensor's address) must recompute, a modified-in-place tensor too.  (b) `_empty_batch` in
    `training_step` on one device: zero loss, no update, the step still counts (HF loop semantics)."""
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments
    d = O.tiny_dims(False)
    model, w = _model(d)
    model.eval()
    b1 = O.synthetic_batch(d, B=2, S_ctx=4, S_tgt=8, seed=31)
    b2 = O.synthetic_batch(d, B=2, S_ctx=4, S_tgt=8, seed=32)
    f1 = b1["batch_features"].cuda()
    b1["batch_features"] = f1
    ref1 = float(model(**b1).loss)
    b2["batch_features"] = b2["batch_features"].cuda()
    ref2 = float(model(**b2).loss)
    assert abs(ref1 - ref2) > 1e-4
    model.prefetch_encoder(f1)
    assert float(model(**b1).loss) == ref1                                  # hit: same object
    model.prefetch_encoder(f1)
    del f1, b1["batch_features"]
    torch.cuda.empty_cache()
    b2n = dict(b2, batch_features=b2["batch_features"].clone())              # same shape, new object (may reuse the freed address)
    assert float(model(**b2n).loss) == ref2                                 # miss -> recomputed, not the stale states of b1
    f3 = b2["batch_features"].clone()
    model.prefetch_encoder(f3)
    f3.mul_(0.5)                                                            # modified after the prefetch: version counter moved
    half = float(model(**dict(b2, batch_features=f3)).loss)
    model.drop_prefetched()
    assert half == float(model(**dict(b2, batch_features=f3.clone())).loss) and half != ref2
    # (b)
    tr = DeSTA25Trainer(model, args=TrainingArguments(learning_rate=1e-3, warmup_steps=0, max_steps=5, logging_steps=1))
    before = model.arena.params.clone()
    loss = tr.training_step({"_empty_batch": True})
    tr.wait_update()
    assert float(loss) == 0.0 and tr.global_step == 1 and tr.optimizer.step_count == 0 and torch.equal(before, model.arena.params)
    tr.training_step(b2)
    tr.wait_update()
    assert tr.global_step == 2 and tr.optimizer.step_count == 1 and not torch.equal(before, model.arena.params)


@pytest.mark.parametrize("flavour", ["llama", "qwen3"])
def test_local_hf_checkpoint_directories_load_like_transformers_wrote_them(tmp_path, flavour):
    """The deployment path: `llm_model_id` / `encoder_model_id` are LOCAL Hugging Face directories (config.json + *.safetensors as
    `save_pretrained` of transformers 5.15 writes them: `rope_parameters`, `model.encoder.*` keys inside a
    WhisperForConditionalGeneration file, bf16 LLM tensors).  `DeSTA25Config` reads the two config.json files, `_load_base_weights`
    the tensors; the result equals the oracle fed with the SAME tensors under the reference's key names."""
    from transformers import LlamaConfig, LlamaForCausalLM, Qwen3Config, Qwen3ForCausalLM, WhisperConfig, WhisperForConditionalGeneration
    from desta.models.modeling_desta25 import CON, DeSTA25AudioModel, DeSTA25Config
    qwen = flavour == "qwen3"
    d = O.tiny_dims(qwen)
    torch.manual_seed(3)
    if qwen:
        lcfg = Qwen3Config(hidden_size=d.llm_h, num_hidden_layers=d.llm_layers, num_attention_heads=d.llm_hq, num_key_value_heads=d.llm_hkv,
                           head_dim=d.llm_hd, intermediate_size=d.llm_inter, vocab_size=d.vocab, rms_norm_eps=d.rms_eps, rope_theta=d.rope_theta,
                           tie_word_embeddings=False, max_position_embeddings=4096)
        llm = Qwen3ForCausalLM(lcfg)
    else:
        f, lo, hi, orig = d.rope_llama3
        lcfg = LlamaConfig(hidden_size=d.llm_h, num_hidden_layers=d.llm_layers, num_attention_heads=d.llm_hq, num_key_value_heads=d.llm_hkv,
                           head_dim=d.llm_hd, intermediate_size=d.llm_inter, vocab_size=d.vocab, rms_norm_eps=d.rms_eps, rope_theta=d.rope_theta,
                           rope_scaling={"rope_type": "llama3", "factor": f, "low_freq_factor": lo, "high_freq_factor": hi,
                                         "original_max_position_embeddings": orig}, tie_word_embeddings=False, max_position_embeddings=4096)
        llm = LlamaForCausalLM(lcfg)
    llm.to(torch.bfloat16).save_pretrained(tmp_path / "llm", safe_serialization=True)
    wcfg = WhisperConfig(num_mel_bins=d.n_mels, d_model=d.enc_d, encoder_layers=d.enc_layers, encoder_attention_heads=d.enc_heads,
                         encoder_ffn_dim=d.enc_ffn, max_source_positions=d.enc_T, decoder_layers=1, decoder_attention_heads=2, decoder_ffn_dim=64,
                         vocab_size=64, max_target_positions=8, pad_token_id=0, bos_token_id=1, eos_token_id=2, decoder_start_token_id=1)
    WhisperForConditionalGeneration(wcfg).save_pretrained(tmp_path / "enc", safe_serialization=True)

    cfg = DeSTA25Config(llm_model_id=str(tmp_path / "llm"), encoder_model_id=str(tmp_path / "enc"), qformer_num_hidden_layers=d.qf_layers,
                        prompt_size=d.prompt_size, qformer_intermediate_size=d.qf_inter, qformer_dropout=0.0)
    lc, ec = cfg.llm_config, cfg.encoder_config
    assert (lc.model_type, lc.hidden_size, lc.num_hidden_layers, lc.num_key_value_heads, lc.head_dim) == (flavour, d.llm_h, d.llm_layers, d.llm_hkv, d.llm_hd)
    assert abs(lc.rope_theta - d.rope_theta) < 1e-3 and bool(lc.rope_scaling) == (not qwen) and lc.qk_norm == qwen
    assert (ec.d_model, ec.encoder_layers, ec.max_source_positions, ec.num_mel_bins) == (d.enc_d, d.enc_layers, d.enc_T, d.n_mels)
    assert cfg.target_layer_ids == [0, 1, 2, 3]
    model = DeSTA25AudioModel(cfg)                                              # weights=None: read from the two directories
    w = DeSTA25AudioModel._load_base_weights(cfg)
    # the encoder for the hot path AND (since round 4) the decoder of the same file: the ASR leg of generate() is built from it
    assert any(k.startswith("perception.whisper.model.encoder.layers.0.") for k in w) and any(k.startswith("perception.whisper.model.decoder.layers.0.") for k in w)
    assert model.asr_decoder is not None and (model.asr_decoder.L, model.asr_decoder.V, model.asr_decoder.Tmax) == (1, 64, 8)
    w = {k: v for k, v in w.items() if ".decoder." not in k}
    w = {k: v.float() for k, v in w.items()}
    w.update({k: v.float().cpu() for k, v in model.state_dict().items()})
    assert all(k.startswith(CON) for k in model.state_dict())
    batch = O.synthetic_batch(d, B=2, S_ctx=5, S_tgt=20, seed=11)
    loss_o, logits_o = O.model_forward(w, d, batch)
    model.eval()
    out = model(**batch)
    m = batch["attention_mask"].bool()
    assert abs(float(out.loss) - float(loss_o)) < 2e-2, (float(out.loss), float(loss_o))
    assert rel_err(out.logits.float().cpu()[m], logits_o[m]) < 3e-2


@pytest.mark.parametrize("name", ["llama", "qwen3"])
def test_dead_row_skip_equals_every_row(name):
    """Training fast path, position-major grid: the last decoder layer's o_proj / MLP / final norm (forward and backward) run on
    the tail of rows from the first target position on, and layer 0's input gradient on the audio rows only.  Both cuts drop
    rows whose values nothing reads (resp. whose gradient is exactly zero): loss and every gradient equal the run over every
    row.  Batch with left padding (different audio starts per sequence) and target spans of different lengths."""
    d = O.tiny_dims(name == "qwen3")
    d.llm_layers = 3
    model, w = _model(d)
    batch = O.synthetic_batch(d, B=3, S_ctx=9, S_tgt=40, seed=5, pad=[0, 7, 2])
    S = batch["input_ids"].shape[1]
    batch["labels"][1, : S - 25] = -100                               # sequence 1: 25 targets, the others 40
    batch["labels"][2, S - 3:] = -100                                 # sequence 2 ends with ignored positions
    res = {}
    model(**batch)                                                    # allocates the activation buffers
    for skip in (True, False):
        model.llm.skip_dead_rows = skip
        for x in (model.llm.xs[-1], model.llm.hb, model.llm.act, model.llm.dxa, model.llm.dxb, model.llm.sv[-1]["xm"], model.llm.sv[-1]["gu"]):
            x.fill_(float("nan"))                                     # anything that reads a skipped row would show
        out = model(**batch)
        assert (model.llm.tail0 == (S - 40 - 1) * 3) == skip, model.llm.tail0
        model.backward()
        res[skip] = (float(out.loss), model.arena.grads.clone())
    assert res[True][0] == res[False][0]
    ga, gb = res[True][1].double(), res[False][1].double()
    assert torch.isfinite(ga).all()
    assert float((ga - gb).norm() / gb.norm()) < 1e-3, float((ga - gb).norm() / gb.norm())
    loss_o, _ = O.model_forward(w, d, batch)
    assert abs(res[True][0] - float(loss_o)) < 2e-2


@pytest.mark.parametrize("name", ["llama", "qwen3"])
def test_swiglu_fusion_equals_separate_kernels(golden_dir, name):
    """Round 4: silu(gate) * up inside the gate|up GEMM epilogue (64-column gate|up blocks, `desta_gemm_desc.act` 2) and its
    backward inside the d(act) GEMM epilogue (act 3) against the unfused path (plain layout + swiglu_fwd / swiglu_bwd kernels):
    same rounding points, so the loss agrees to bf16 noise of a handful of elements and every gradient to 2e-3; both stay on the
    reference's golden.  (A/B switch: `model.llm.fuse_swiglu`, bench.py --no-swiglu-fusion.)"""
    d = O.tiny_dims(name == "qwen3")
    g, batch = golden_batch(golden_dir, name)
    model, w = _model(d)
    assert all("wgu_b" in ly for ly in model.llm.layers)
    res = {}
    for fuse in (True, False):
        model.llm.fuse_swiglu = fuse
        out = model(**batch)
        assert all(s["gu_blocked"] == fuse for s in model.llm.sv)
        model.backward()
        res[fuse] = (float(out.loss), model.arena.grads.clone())
    assert abs(res[True][0] - res[False][0]) < 2e-4, (res[True][0], res[False][0])
    a, b = res[True][1].double(), res[False][1].double()
    assert float((a - b).norm() / b.norm()) < 2e-3, float((a - b).norm() / b.norm())
    assert abs(res[True][0] - float(g["loss"])) < 2e-3
    # eval forward (batch-major grid, logits kept) takes the fused projection too and matches the golden logits
    model.eval()
    model.llm.fuse_swiglu = True
    out = model(**batch, keep_logits=True)
    m = g["attention_mask"].bool()
    assert rel_err(out.logits.float().cpu()[m], g["logits"][m]) < 2e-2
