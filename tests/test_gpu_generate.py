"""GPU parity of `_generate_step` (KV-cached greedy decoding on the HIP path) against
  (1) the tokens the REFERENCE's own `_generate_step` produced (tests/golden/ref_tiny_*.safetensors: gen_ids / gen_ids_eos),
  (2) the cache-free oracle (oracle.greedy_generate) step by step under teacher forcing.
The product computes in bf16, the reference in fp32: per-step logits must agree to 3e-2 relative L2, and every
greedy choice of the product must be (near-)optimal under the fp32 logits (argmax flips only between near-ties)."""
import math
import os

import pytest
import torch

import desta_oracle as O
from helpers import cfg_from_dims, golden_batch, rel_err

pytestmark = pytest.mark.gpu


def _model(d, seed=7):
    from desta.models.modeling_desta25 import DeSTA25AudioModel
    w = O.init_weights(d, seed=seed)
    return DeSTA25AudioModel(cfg_from_dims(d), weights=w), w


def _gen_inputs(g, batch):
    n_ctx = int(g["gen_ctx_len"])
    return {"context_input_ids": batch["input_ids"][:, :n_ctx], "context_attention_mask": batch["attention_mask"][:, :n_ctx],
            "context_batch_start_positions": batch["batch_start_positions"], "batch_features": batch["batch_features"],
            "batch_transcription_ids": batch["batch_transcription_ids"]}


def _oracle_logits(w, d, g, batch, forced):
    n_ctx = int(g["gen_ctx_len"])
    with torch.no_grad():
        af = O.perception(w, d, batch["batch_features"])
        x = O.embed_splice(w, d, batch["input_ids"][:, :n_ctx], af, batch["batch_transcription_ids"], batch["batch_start_positions"])
        return O.greedy_generate(w, d, x, batch["attention_mask"][:, :n_ctx], forced.shape[1], 0, forced_tokens=forced)[1]


@pytest.mark.parametrize("name", ["llama", "qwen3"])
def test_generate_step_vs_reference_golden(golden_dir, name):
    d = O.tiny_dims(name == "qwen3")
    g, batch = golden_batch(golden_dir, name)
    model, w = _model(d)
    inputs = _gen_inputs(g, batch)
    ref_ids = g["gen_ids"]
    # (a) teacher-forced on the reference's tokens: the cached decode reproduces the fp32 logits of every step
    ids_f, logits = model._generate_step(inputs, pad_token_id=0, max_new_tokens=10, do_sample=False,
                                         forced_tokens=ref_ids, collect_logits=True)
    assert ids_f.cpu().tolist() == ref_ids.tolist()
    lo = _oracle_logits(w, d, g, batch, ref_ids)
    assert logits.shape == lo.shape
    for t in range(10):
        e = rel_err(logits[t].float(), lo[t])
        assert e < 3e-2, (t, e)
    # the product's own argmax at every step is within bf16 noise of the fp32 optimum
    pick = logits.float().cpu().argmax(-1)                                   # [T, B]
    gap = lo.max(-1).values - lo.gather(-1, pick.unsqueeze(-1)).squeeze(-1)
    spread = lo.std(-1)
    assert float((gap / spread).max()) < 0.1, (gap / spread)
    agree = float((pick == lo.argmax(-1)).float().mean())
    assert agree >= 0.9, agree
    # (b) free running: a bf16 near-tie may pick another token than the fp32 reference and the continuations then
    # differ legitimately, so check the product's OWN path: every token it chose is (near-)optimal under the fp32
    # logits of the oracle run on that same prefix, and the first token equals the reference's
    ids = model._generate_step(inputs, pad_token_id=0, max_new_tokens=10, do_sample=False).cpu()
    assert ids.shape == ref_ids.shape
    print("free-running token agreement with the reference:", float((ids == ref_ids).float().mean()))
    assert (ids[:, 0] == ref_ids[:, 0]).all()
    lo2 = _oracle_logits(w, d, g, batch, ids)                                # [T, B, V]
    gap2 = lo2.max(-1).values - lo2.gather(-1, ids.t().unsqueeze(-1)).squeeze(-1)
    assert float((gap2 / lo2.std(-1)).max()) < 0.1, gap2 / lo2.std(-1)
    # (c) EOS rule: finished rows are padded, other rows continue (teacher-forced so the EOS step is hit exactly)
    eos = int(g["gen_eos_id"])
    ids_e = model._generate_step(inputs, pad_token_id=0, max_new_tokens=10, do_sample=False, eos_token_id=eos,
                                 forced_tokens=ref_ids).cpu()
    assert ids_e.tolist() == g["gen_ids_eos"].tolist()


def test_generate_early_stop_and_no_audio():
    """All rows hit EOS -> output is trimmed like HF's stopping criteria; a text-only batch works too."""
    d = O.tiny_dims(False)
    model, w = _model(d)
    gen = torch.Generator().manual_seed(3)
    ids = torch.randint(3, d.vocab, (3, 21), generator=gen)
    am = torch.ones(3, 21, dtype=torch.long)
    am[1, :6] = 0
    ids[1, :6] = 0
    inputs = {"context_input_ids": ids, "context_attention_mask": am, "context_batch_start_positions": [],
              "batch_features": None, "batch_transcription_ids": []}
    with torch.no_grad():
        x = O.embed_splice(w, d, ids, None, [], [])
        ref, ref_logits = O.greedy_generate(w, d, x, am, 24, 0)
    out, logits = model._generate_step(inputs, pad_token_id=0, max_new_tokens=24, do_sample=False, forced_tokens=ref, collect_logits=True)
    assert out.cpu().tolist() == ref.tolist()
    for t in range(24):                                  # covers prompt lengths / cache lengths across a 32-key tile edge
        assert rel_err(logits[t].float(), ref_logits[t]) < 3e-2, t
    # every row's first token is its EOS -> one column
    first = ref[:, 0].tolist()
    out1 = model._generate_step(inputs, pad_token_id=0, max_new_tokens=24, do_sample=False, eos_token_id=first, forced_tokens=ref)
    ref1, _ = O.greedy_generate(w, d, x, am, 24, 0, eos_token_ids=first, forced_tokens=ref)
    assert out1.cpu().tolist() == ref1.tolist() and out1.shape[1] == 1
    with pytest.raises(RuntimeError, match="needs a tokenizer"):           # chat-level generate(): the front end is injected
        model.generate([{"role": "user", "content": "hi"}])
    # ... and with one, a text-only conversation goes through the KV-cached decoder (no audio: plain LLM generation)
    from helpers import ToyTokenizer
    model._setup_generation(tokenizer=ToyTokenizer(vocab_size=d.vocab), processor=object())
    out = model.generate([[{"role": "user", "content": "hi there"}], [{"role": "user", "content": "a somewhat longer question , please"}]],
                         do_sample=False, max_new_tokens=5)
    assert len(out.text) == 2 and len(out.generated_ids) == 2 and out.audios == [] and all(len(r) <= 5 for r in out.generated_ids)


def test_argmax_kernel():
    from desta import _hip as H
    g = torch.Generator().manual_seed(0)
    x = torch.randn(7, 1000, generator=g).to(torch.bfloat16).cuda()
    x[2, 17] = x[2, 900] = 9.0                           # tie: first index wins, like torch.argmax
    x[5, 999] = 11.0
    out = torch.zeros(7, dtype=torch.int64, device="cuda")
    H.argmax_bf16(x, 1000, 7, 1000, out)
    assert out.cpu().tolist() == x.float().cpu().argmax(-1).tolist()
    assert int(out[2]) == 17 and int(out[5]) == 999
    H.argmax_bf16(x, 1000, 7, 900, out)                 # only the first `cols` entries are searched
    assert int(out[5]) != 999 and int(out[2]) == 17


@pytest.mark.parametrize("V,temp,top_p", [(1000, 0.7, 0.9), (128256, 0.7, 0.9), (151936, 1.3, 0.5), (512, 1.0, 1.0), (4099, 0.3, 0.999)])
def test_top_p_kept_set_matches_hf_warpers(V, temp, top_p):
    """The kept set of desta_sample_top_p_bf16 == TemperatureLogitsWarper -> TopPLogitsWarper of transformers on the same
    (bf16-valued) logits.  Allowed differences: tokens whose cumulative mass is within 2e-5 of the cut-off (fp32 summation
    order) and tokens that tie with the boundary logit (torch.sort leaves their order unspecified; we keep all of them)."""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopPLogitsWarper
    from desta import _hip as H
    g = torch.Generator().manual_seed(V)
    rows = 5
    logits = (torch.randn(rows, V, generator=g) * 3.0).to(torch.bfloat16)
    logits[1, :7] = 9.0                                   # a tie group at the top
    ld = V + 8 - V % 8
    buf = torch.zeros(rows, ld, dtype=torch.bfloat16, device="cuda")
    buf[:, :V] = logits.cuda()
    out = torch.zeros(rows, dtype=torch.int64, device="cuda")
    mask = torch.zeros(rows, V, dtype=torch.uint8, device="cuda")
    H.sample_top_p(buf, ld, rows, V, temp, top_p, 1234, 0, out, keep_mask=mask)
    kept = mask.cpu().bool()
    scores = TemperatureLogitsWarper(temp)(None, logits.float())
    if top_p < 1.0:
        scores = TopPLogitsWarper(top_p)(None, scores)
    ref = torch.isfinite(scores)
    probs = torch.softmax(logits.float() / temp, -1)
    for r in range(rows):
        diff = (kept[r] != ref[r]).nonzero().flatten()
        if diff.numel():
            sp, _ = torch.sort(probs[r])
            cum = torch.cumsum(sp.double(), 0)
            for i in diff.tolist():
                c_i = float(cum[int((sp <= probs[r, i]).sum()) - 1])          # mass of tokens not more probable than i
                boundary_tie = bool((probs[r][ref[r]].min() == probs[r, i]) or (probs[r][kept[r]].min() == probs[r, i]))
                assert abs(c_i - (1.0 - top_p)) < 2e-5 or boundary_tie, (r, i, c_i)
        assert kept[r, int(logits[r].float().argmax())]                      # min_tokens_to_keep = 1
        assert kept[r, int(out[r])]                                          # the sample is one of the kept tokens
    # (every differing token was justified above; with 128k bf16 logits the boundary tie group alone holds ~20 tokens)
    assert int(kept.sum()) > 0 and int((ref & ~kept).sum()) <= rows


def test_top_p_sampling_distribution():
    """Frequencies of 20000 draws (different step counters) match the renormalised kept probabilities."""
    from desta import _hip as H
    V, n = 64, 20000
    g = torch.Generator().manual_seed(2)
    logits = (torch.randn(1, V, generator=g) * 2).to(torch.bfloat16)
    probs = torch.softmax(logits.float() / 0.8, -1)[0]
    rows = 8
    buf = logits.cuda().repeat(rows, 1).contiguous()
    mask = torch.zeros(rows, V, dtype=torch.uint8, device="cuda")
    out = torch.zeros(rows, dtype=torch.int64, device="cuda")
    counts = torch.zeros(V)
    outs = []
    for step in range(n // rows):
        H.sample_top_p(buf, V, rows, V, 0.8, 0.9, 99, step, out, keep_mask=mask)
        outs.append(out.clone())
    toks = torch.cat(outs).cpu()
    counts = torch.bincount(toks, minlength=V).float()
    kept = mask[0].cpu().bool()
    assert counts[~kept].sum() == 0
    q = torch.where(kept, probs, torch.zeros_like(probs))
    q = q / q.sum()
    freq = counts / counts.sum()
    # 5-sigma binomial bound per token
    sigma = torch.sqrt(q * (1 - q) / n)
    assert bool(((freq - q).abs() <= 5 * sigma + 1e-4).all()), (freq - q).abs().max()
    # reproducible: same (seed, step, row) -> same token; another seed -> another sequence
    o1, o2, o3 = (torch.zeros(rows, dtype=torch.int64, device="cuda") for _ in range(3))
    H.sample_top_p(buf, V, rows, V, 0.8, 0.9, 99, 5, o1)
    H.sample_top_p(buf, V, rows, V, 0.8, 0.9, 99, 5, o2)
    assert torch.equal(o1, o2) and torch.equal(o1, outs[5])
    assert len(set(toks[:64].tolist())) > 4


def test_generate_step_sampling_end_to_end(golden_dir):
    """do_sample=True through _generate_step: reproducible per seed, every sampled token inside the top-p set of the fp32
    oracle logits for the product's own prefix (up to bf16 noise at the set boundary), top_p -> 0 degenerates to greedy."""
    d = O.tiny_dims(False)
    g, batch = golden_batch(golden_dir, "llama")
    model, w = _model(d)
    inputs = _gen_inputs(g, batch)
    a = model._generate_step(inputs, pad_token_id=0, max_new_tokens=10, do_sample=True, temperature=0.7, top_p=0.9, seed=3).cpu()
    b = model._generate_step(inputs, pad_token_id=0, max_new_tokens=10, do_sample=True, temperature=0.7, top_p=0.9, seed=3).cpu()
    c = model._generate_step(inputs, pad_token_id=0, max_new_tokens=10, do_sample=True, temperature=0.7, top_p=0.9, seed=4).cpu()
    assert torch.equal(a, b) and not torch.equal(a, c)
    lo = _oracle_logits(w, d, g, batch, a)                                   # [T, B, V] fp32 logits on the sampled prefix
    p = torch.softmax(lo / 0.7, -1)
    sp, _ = torch.sort(p, dim=-1)
    cum = torch.cumsum(sp, -1)
    for t in range(a.shape[1]):
        for r in range(a.shape[0]):
            pi = p[t, r, a[r, t]]
            c_i = float(cum[t, r, int((sp[t, r] <= pi).sum()) - 1])
            assert c_i > (1.0 - 0.9) - 2e-2, (t, r, c_i)                      # inside the nucleus (bf16 logits move the edge slightly)
    # top_p -> 0 keeps only the most probable logit value: every sampled token attains the row maximum of the step's
    # logits (bf16 logits can TIE — then greedy takes the first index, sampling any member of the tie group)
    nearly, lg = model._generate_step(inputs, pad_token_id=0, max_new_tokens=6, do_sample=True, temperature=0.7, top_p=1e-6, seed=1,
                                      collect_logits=True)
    lg = lg.float()
    picked = lg.gather(-1, nearly.t().unsqueeze(-1)).squeeze(-1)
    assert torch.equal(picked, lg.max(-1).values)


def test_trainer_evaluate_loss_ppl_and_predictions(golden_dir):
    """DeSTA25Trainer.evaluate (reference desta_trainer.py:104-189): eval-mode loss / ppl == the oracle's forward loss
    (no dropout even with qformer_dropout configured), predictions == _generate_step on the batch's context part;
    with a tokenizer-like `processing_class` the decoded strings are recorded under the reference's keys."""
    from desta.models.modeling_desta25 import DeSTA25AudioModel
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments
    d = O.tiny_dims(False)
    g, batch = golden_batch(golden_dir, "llama")
    w = O.init_weights(d, seed=7)
    model = DeSTA25AudioModel(cfg_from_dims(d, dropout=0.1), weights=w)
    full = dict(batch)
    full.update(_gen_inputs(g, batch))
    full["metadata"] = [{"id": "a"}, {"id": "b"}]
    tr = DeSTA25Trainer(model, args=TrainingArguments(max_steps=10))
    m = tr.evaluate([full, {"_empty_batch": True}, full], generation_kwargs=dict(max_new_tokens=10, do_sample=False))
    assert abs(m["eval_loss"] - float(g["loss"])) < 2e-2 and abs(m["eval_ppl"] - math.exp(float(g["loss"]))) < 2e-2 * math.exp(float(g["loss"]))
    assert model.training                                        # mode restored
    ids = model.eval()._generate_step(full, pad_token_id=0, max_new_tokens=10, do_sample=False).cpu().tolist()
    model.train()
    assert len(tr.prediction_step_outputs) == 4
    assert [o["prediction_ids"] for o in tr.prediction_step_outputs[:2]] == ids and tr.prediction_step_outputs[0]["id"] == "a"

    class Tok:                                                   # minimal tokenizer surface used by _predict_step
        eos_token_id = 2

        def batch_decode(self, x, skip_special_tokens=False):
            return [" ".join(str(int(t)) for t in row) for row in x]
    tr2 = DeSTA25Trainer(model, args=TrainingArguments(max_steps=10), processing_class=Tok())
    tr2.evaluate([full], generation_kwargs=dict(max_new_tokens=4, do_sample=False))
    o = tr2.prediction_step_outputs[0]
    assert set(o) >= {"context", "prediction", "label", "id"} and len(o["prediction"].split()) == 4


def _asr_model(golden_dir, gen_cfg=None):
    from helpers import ASR_DIMS, ASR_GEN_CFG, asr_weights, cfg_from_dims
    from safetensors.torch import load_file
    from desta.models.modeling_desta25 import DeSTA25AudioModel
    d = O.tiny_dims(False)
    w = {**O.init_weights(d, seed=7), **asr_weights(d, seed=5)}
    cfg = cfg_from_dims(d, whisper_generation_config=dict(gen_cfg or ASR_GEN_CFG))
    e = cfg.encoder_config
    e.decoder_layers, e.vocab_size, e.max_target_positions = ASR_DIMS["decoder_layers"], ASR_DIMS["vocab_size"], ASR_DIMS["max_target_positions"]
    e.decoder_attention_heads, e.decoder_ffn_dim = d.enc_heads, d.enc_ffn
    g = load_file(os.path.join(golden_dir, "ref_asr_tiny.safetensors"))
    return DeSTA25AudioModel(cfg, weights=w), g, d


def test_whisper_asr_decoder_vs_reference_golden(golden_dir):
    """The ASR leg of generate() (reference: `self.perception.whisper.generate(input_features, attention_mask=None, max_new_tokens=128)`,
    modeling_desta25.py:1580-1590) on the device: frozen encoder + encoder.layer_norm -> KV-cached greedy decoder.  Golden
    (tests/golden/ref_asr_tiny.safetensors): the model's own forward under the generate rules — init tokens [start, detected language,
    no-timestamps], suppress / begin-suppress lists, EOS stop — which the golden script checks token for token against
    WhisperForConditionalGeneration.generate on a short budget.  Asserted: the init tokens (language detection included), the raw
    logits of every step on the golden's own prefix (teacher forcing), the free-running tokens, EOS stop + padding."""
    from helpers import ASR_GEN_CFG
    model, g, d = _asr_model(golden_dir)
    assert model.asr_decoder is not None
    n_init = int(g["n_init"])
    seq = g["sequences"]
    N = seq.shape[1] - n_init
    feats = g["batch_features"]
    ids, logits, init = model.asr_decoder.generate(model.encoder, feats, max_new_tokens=N, collect_logits=True, forced_tokens=seq[:, n_init:])
    assert init.cpu().tolist() == seq[:, :n_init].tolist()                        # [start, <|de|> by detection, no-timestamps]; no task token (language unset)
    assert ids.cpu().tolist() == seq[:, n_init:].tolist()
    e = rel_err(logits.float().cpu(), g["logits"])
    print("asr decoder: teacher-forced logits rel-L2", e)
    assert e < 2e-2, e                                                            # bf16 operands / fp32 residual stream vs the fp32 reference
    # free running: every token whose golden margin over the runner-up is comfortable must come out the same
    free = model.asr_decoder.generate(model.encoder, feats, max_new_tokens=N).cpu()
    gl = g["logits"].clone()                                                       # [N, B, V] raw; apply the suppress rules for the margin
    gl[:, :, ASR_GEN_CFG["suppress_tokens"]] = -1e30
    gl[0][:, ASR_GEN_CFG["begin_suppress_tokens"]] = -1e30
    top2 = gl.topk(2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1]).t()                                     # [B, N]
    agree = 0
    for b in range(free.shape[0]):
        for k in range(min(N, free.shape[1])):
            if int(free[b, k]) != int(seq[b, n_init + k]):
                assert float(margin[b, k]) < 0.35, (b, k, float(margin[b, k]))     # only a near-tie may flip; the prefix differs from here on
                break
            agree += 1
    assert agree >= 0.8 * free.numel(), (agree, free.tolist(), seq[:, n_init:].tolist())
    # EOS variant: stop, pad with pad_token_id (= eos here), trim behind the last EOS
    eos = int(g["eos_id"])
    gc2 = {**ASR_GEN_CFG, "eos_token_id": eos, "pad_token_id": eos, "bos_token_id": eos, "begin_suppress_tokens": [7, eos]}
    model2, _, _ = _asr_model(golden_dir, gc2)
    out = model2.asr_decoder.generate(model2.encoder, feats, max_new_tokens=N).cpu()
    want = g["sequences_eos"][:, n_init:]
    assert out.shape[1] <= want.shape[1]
    for b in range(out.shape[0]):
        row = want[b].tolist()
        if eos in row:                                                             # rows that finish: identical up to and including EOS, padding after it
            k = row.index(eos) + 1
            assert out[b, :k].tolist() == row[:k] and all(int(t) == eos for t in out[b, k:]), (b, out[b].tolist(), row)


def test_chat_generate_transcribes_speech_without_text(golden_dir, tmp_path):
    """Chat-level `generate(messages)` with an audio that has speech and NO 'text': the Whisper decoder of the checkpoint transcribes it
    (no injected `asr`), the transcription's tokens follow the audio features in the prompt (modeling_desta25.py:1562-1600)."""
    import numpy as np
    from helpers import ToyTokenizer
    from desta.utils.audio import HipLogMelProcessor
    model, g, d = _asr_model(golden_dir)
    d_T = d.enc_T

    class AsrTok:
        def batch_decode(self, ids, skip_special_tokens=True):
            return [" ".join(f"w{int(t)}" for t in row if int(t) > 15) for row in ids]
    # a 1-s WAVE file; the tiny encoder takes 2 * 96 mel frames, so the processor hands over the golden's features instead of a 3000-frame log-mel
    import wave
    p = tmp_path / "speech.wav"
    with wave.open(str(p), "wb") as wv:
        wv.setnchannels(1); wv.setsampwidth(2); wv.setframerate(16000)
        wv.writeframes((0.1 * np.random.default_rng(0).standard_normal(16000) * 32767).astype("<i2").tobytes())

    class Proc:
        def __call__(self, waves, sampling_rate=None, return_tensors=None):
            import types
            return types.SimpleNamespace(input_features=g["batch_features"][:len(waves)].cuda())
    model._setup_generation(tokenizer=ToyTokenizer(vocab_size=d.vocab), processor=Proc(), vad=lambda w: True, asr_tokenizer=AsrTok())
    assert model.asr is not None                                                   # built from the decoder, not injected
    out = model.generate([{"role": "user", "content": "What is said? <|AUDIO|>", "audios": [{"audio": str(p), "text": None}]}],
                         do_sample=False, max_new_tokens=4)
    text = out.audios[0][1]
    want = " ".join(f"w{int(t)}" for t in model._last_asr_ids[0].tolist() if int(t) > 15)
    assert text == want.strip() and len(text) > 0
    n_tr = len(ToyTokenizer().tokenize(text))
    assert model._last_generate_inputs["batch_transcription_ids"][0].shape == (1, n_tr)
    assert len(out.generated_ids[0]) == 4
