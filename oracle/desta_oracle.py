"""CPU oracle for the DeSTA2.5-Audio training step (TEST INFRASTRUCTURE ONLY).

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``desta2.5-audio_amd/desta``) never imports anything under ``oracle/`` and fails loudly when
``libdesta_hip.so`` is missing.

It is a plain-PyTorch fp32 restatement of the reference hot path, written functionally over a
flat ``{name: tensor}`` weight dict that uses the reference's own state-dict key names:

* log-mel ............ transformers ``WhisperFeatureExtractor._torch_extract_fbank_features``
                       (``TF:models/whisper/feature_extraction_whisper.py:135-168``) as called at
                       ``desta/trainer/data/simple_dataset.py:239-243``; slaney filter bank
                       ``TF:audio_utils.py:638-729``.
* Whisper stem+layers  ``desta/models/modeling_desta25.py:551-585`` →
                       ``TF:models/whisper/modeling_whisper.py:241-413`` (4.x tuple semantics:
                       the WHOLE hidden-state tensor is carried, hazard H1; final encoder
                       layer_norm is never applied, hazard H2).
* Q-Former ........... ``modeling_desta25.py:126-176, 587-598`` → BERT decoder layer
                       ``TF:models/bert/modeling_bert.py:354-448`` (post-LN, bidirectional
                       self-attention, unmasked cross-attention, eps 1e-12, GELU-erf).
* tap mix + projector  ``modeling_desta25.py:600-606``.
* embed + splice ..... ``modeling_desta25.py:1009-1041``.
* causal LM .......... ``TF:models/llama/modeling_llama.py:53-492`` (llama3 rope scaling
                       ``TF:modeling_rope_utils.py``), Qwen3 per-head q/k RMSNorm
                       ``TF:models/qwen3/modeling_qwen3.py:237-257``; loss
                       ``TF:loss/loss_utils.py:49-71``.
* LoRA (use_lora) .... ``modeling_desta25.py:720-729``: ``peft.LoraConfig(r=16, lora_alpha=16, lora_dropout=0.1,
                       target_modules=[q_proj, k_proj, v_proj])``.  ``peft`` is a third-party dependency that is neither
                       vendored under /root/reference nor installed here (reference pin: ``peft`` without a version,
                       ``setup.py``): PARITY UNPINNED for this branch.  Restated from the published LoRA layer:
                       ``y = base(x) + lora_B(lora_A(dropout(x))) * (alpha / r)``, A ~ kaiming_uniform(a=sqrt 5), B = 0,
                       adapter weights fp32 (peft casts bf16 adapters up), keys ``<module>.lora_{A,B}.default.weight``.
* clip + Adafactor ... ``TF:trainer.py:1778-1797``; ``TF:optimization.py:1203-1294`` with the
                       Trainer kwargs ``scale_parameter=False, relative_step=False``.

Parity pin: ``tests/test_oracle_pin.py`` checks every stage of this file against the installed
transformers blocks (weights copied) and against goldens produced by running the reference's own
``QformerConnector`` / ``WhisperPerception.forward_whisper`` / ``_prepare_inputs_for_llm`` code
(``tests/golden/make_golden_from_reference.py``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BF16 = torch.bfloat16

# ----------------------------------------------------------------------------- precision policy
# fp32 (default): every op in fp32 — the goldens produced by the reference's code are fp32 as well.
# autocast_bf16(): the CASTING POLICY of ``torch.autocast("cuda", dtype=torch.bfloat16)`` restated by hand, which is
# what HF Trainer(bf16=True) wraps ``compute_loss`` in (``TF:trainer.py:1930``, hazard H11): linear / matmul / conv
# run on bf16 operands and return bf16, layer_norm and softmax run in fp32 and return fp32, everything else follows
# type promotion (fp32 + bf16 -> fp32).  LLM weights are stored in bf16 (``modeling_desta25.py:715``), so its whole
# residual stream is bf16; Whisper / Q-Former weights are fp32, so their residual streams stay fp32.  CPU autocast has
# a different op list (softmax stays bf16 there), hence the explicit restatement instead of torch.autocast("cpu").
_AC = False


class autocast_bf16:
    def __enter__(self):
        global _AC
        self._prev, _AC = _AC, True
        return self

    def __exit__(self, *exc):
        global _AC
        _AC = self._prev
        return False


def _lin(x, w, b=None):
    if _AC:
        return F.linear(x.to(BF16), w.to(BF16), None if b is None else b.to(BF16))
    return F.linear(x, w, b)


def _mm(a, b):
    if _AC:
        return a.to(BF16) @ b.to(BF16)
    return a @ b


def _conv1d(x, w, b, **kw):
    if _AC:
        return F.conv1d(x.to(BF16), w.to(BF16), b.to(BF16), **kw)
    return F.conv1d(x, w, b, **kw)


def _ln(x, n, g, b, eps):
    return F.layer_norm(x.float(), (n,), g.float(), b.float(), eps)


def _softmax(s):
    return torch.softmax(s.float(), dim=-1)


# ----------------------------------------------------------------------------- dims
@dataclass
class Dims:
    """Shape/config record shared by the oracle and the tests (no model keys leak to bench)."""
    # Whisper encoder
    n_mels: int = 128
    enc_d: int = 1280
    enc_layers: int = 32
    enc_heads: int = 20
    enc_ffn: int = 5120
    enc_T: int = 1500                      # max_source_positions; mel frames = 2*enc_T
    taps: Tuple[int, ...] = (7, 15, 23, 31)
    # Q-Former
    qf_layers: int = 6
    qf_inter: int = 3072                   # BertConfig() default, never overridden (modeling_desta25.py:156-162)
    prompt_size: int = 64
    # LLM
    llm_h: int = 4096
    llm_layers: int = 32
    llm_hq: int = 32
    llm_hkv: int = 8
    llm_hd: int = 128
    llm_inter: int = 14336
    vocab: int = 128256
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_llama3: Optional[Tuple[float, float, float, int]] = (8.0, 1.0, 4.0, 8192)  # factor, low, high, orig ctx
    qk_norm: bool = False                  # Qwen3
    tie_embeddings: bool = False
    # LoRA on the decoder's q/k/v projections (modeling_desta25.py:720-729); 0 = off
    lora_r: int = 0
    lora_alpha: float = 16.0

    @property
    def qf_heads(self) -> int:             # num_attention_heads = encoder heads (modeling_desta25.py:158)
        return self.enc_heads


def tiny_dims(qwen3: bool = False) -> Dims:
    """Debug-config stand-in: whisper-tiny-like encoder, tiny local-config causal LM, Q-Former 2L."""
    return Dims(n_mels=80, enc_d=128, enc_layers=4, enc_heads=2, enc_ffn=256, enc_T=96,
                taps=(0, 1, 2, 3), qf_layers=2, qf_inter=192, prompt_size=64,
                llm_h=256, llm_layers=2, llm_hq=4, llm_hkv=2, llm_hd=128 if qwen3 else 64,
                llm_inter=512, vocab=512, rms_eps=1e-6 if qwen3 else 1e-5,
                rope_theta=1e6 if qwen3 else 500000.0,
                rope_llama3=None if qwen3 else (8.0, 1.0, 4.0, 64),
                qk_norm=qwen3, tie_embeddings=False)


def deep_dims(qwen3: bool = False) -> Dims:
    """Tiny WIDTH at the reference's real DEPTH: 32 encoder layers tapped at 7/15/23/31 (whisper-large-v3[-turbo],
    modeling_desta25.py:140-143), Q-Former 6L (every shipped *_Qformer6L.yaml), 32 (Llama-3.1-8B) or 36 (Qwen3-4B/8B)
    decoder layers — pins how the bf16-vs-fp32 error grows with depth."""
    d = tiny_dims(qwen3)
    d.enc_layers, d.taps, d.qf_layers, d.llm_layers = 32, (7, 15, 23, 31), 6, (36 if qwen3 else 32)
    return d


def tied_dims() -> Dims:
    """Qwen3-4B-like head geometry at tiny width: Hq*hd (512) != hidden (320), q/k-norm, TIED lm_head
    (examples/train/config/desta25_qwen3-4B_Qformer6L.yaml; modeling_desta25.py:638 of the product)."""
    d = tiny_dims(True)
    d.llm_h, d.llm_hq, d.llm_hkv, d.llm_hd, d.tie_embeddings = 320, 4, 2, 128, True
    return d


ENC = "perception.whisper.model.encoder."
CON = "perception.connector."
LLM = "llm_model."


def trainable_names(d: Dims) -> List[str]:
    """Names of the trainable (connector) tensors, in the reference's named_parameters order
    (``modeling_desta25.py:148-168``; ``configure_trainable_parameters`` ``:1439-1463``)."""
    names = [f"{CON}layer_weights"]
    names += [f"{CON}layer_prompts.{j}" for j in range(len(d.taps))]
    for i in range(d.qf_layers):
        p = f"{CON}qformer.layer.{i}."
        for blk in ("attention", "crossattention"):
            for lin in ("self.query", "self.key", "self.value", "output.dense", "output.LayerNorm"):
                names += [f"{p}{blk}.{lin}.weight", f"{p}{blk}.{lin}.bias"]
        for lin in ("intermediate.dense", "output.dense", "output.LayerNorm"):
            names += [f"{p}{lin}.weight", f"{p}{lin}.bias"]
    names += [f"{CON}proj.0.weight", f"{CON}proj.0.bias", f"{CON}proj.1.weight", f"{CON}proj.1.bias"]
    return lora_names(d) + names            # llm_model is registered before perception (modeling_desta25.py:713, 732)


def lora_names(d: Dims) -> List[str]:
    """peft's parameter names under ``get_peft_model(...).base_model.model`` (``modeling_desta25.py:729``)."""
    if not d.lora_r:
        return []
    return [f"{LLM}model.layers.{i}.self_attn.{m}_proj.lora_{ab}.default.weight"
            for i in range(d.llm_layers) for m in "qkv" for ab in "AB"]


def init_weights(d: Dims, seed: int = 0, scale: float = 1.0) -> Dict[str, Tensor]:
    """Seeded random weights at the shapes of ``d`` (there are no checkpoints offline).
    Connector follows torch default Linear init / randn prompts / zero mix weights (hazard H4)."""
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, Tensor] = {}

    def lin(name, out_f, in_f, bias=True, std=None):
        k = 1.0 / math.sqrt(in_f) if std is None else std
        w[name + ".weight"] = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * k * scale
        if bias:
            w[name + ".bias"] = (torch.rand(out_f, generator=g) * 2 - 1) * k

    def ln(name, n, bias=True):
        w[name + ".weight"] = 1.0 + 0.1 * torch.randn(n, generator=g)
        if bias:
            w[name + ".bias"] = 0.1 * torch.randn(n, generator=g)

    # Whisper encoder
    w[ENC + "conv1.weight"] = torch.randn(d.enc_d, d.n_mels, 3, generator=g) / math.sqrt(3 * d.n_mels)
    w[ENC + "conv1.bias"] = 0.1 * torch.randn(d.enc_d, generator=g)
    w[ENC + "conv2.weight"] = torch.randn(d.enc_d, d.enc_d, 3, generator=g) / math.sqrt(3 * d.enc_d)
    w[ENC + "conv2.bias"] = 0.1 * torch.randn(d.enc_d, generator=g)
    w[ENC + "embed_positions.weight"] = 0.1 * torch.randn(d.enc_T, d.enc_d, generator=g)
    for i in range(d.enc_layers):
        p = f"{ENC}layers.{i}."
        lin(p + "self_attn.q_proj", d.enc_d, d.enc_d)
        lin(p + "self_attn.k_proj", d.enc_d, d.enc_d, bias=False)
        lin(p + "self_attn.v_proj", d.enc_d, d.enc_d)
        lin(p + "self_attn.out_proj", d.enc_d, d.enc_d)
        ln(p + "self_attn_layer_norm", d.enc_d)
        lin(p + "fc1", d.enc_ffn, d.enc_d)
        lin(p + "fc2", d.enc_d, d.enc_ffn)
        ln(p + "final_layer_norm", d.enc_d)
    # connector
    for j in range(len(d.taps)):
        w[f"{CON}layer_prompts.{j}"] = torch.randn(1, d.prompt_size, d.enc_d, generator=g)
    w[f"{CON}layer_weights"] = 0.3 * torch.randn(d.prompt_size, len(d.taps), generator=g)
    for i in range(d.qf_layers):
        p = f"{CON}qformer.layer.{i}."
        for blk in ("attention", "crossattention"):
            lin(p + blk + ".self.query", d.enc_d, d.enc_d)
            lin(p + blk + ".self.key", d.enc_d, d.enc_d)
            lin(p + blk + ".self.value", d.enc_d, d.enc_d)
            lin(p + blk + ".output.dense", d.enc_d, d.enc_d)
            ln(p + blk + ".output.LayerNorm", d.enc_d)
        lin(p + "intermediate.dense", d.qf_inter, d.enc_d)
        lin(p + "output.dense", d.enc_d, d.qf_inter)
        ln(p + "output.LayerNorm", d.enc_d)
    ln(CON + "proj.0", d.enc_d)
    lin(CON + "proj.1", d.llm_h, d.enc_d)
    # LLM
    # tied: the same matrix is the lm_head, so keep the logits O(1) (real tied checkpoints have std ~0.02-0.05)
    w[LLM + "model.embed_tokens.weight"] = (0.06 if d.tie_embeddings else 0.5) * torch.randn(d.vocab, d.llm_h, generator=g)
    for i in range(d.llm_layers):
        p = f"{LLM}model.layers.{i}."
        ln(p + "input_layernorm", d.llm_h, bias=False)
        ln(p + "post_attention_layernorm", d.llm_h, bias=False)
        lin(p + "self_attn.q_proj", d.llm_hq * d.llm_hd, d.llm_h, bias=False)
        lin(p + "self_attn.k_proj", d.llm_hkv * d.llm_hd, d.llm_h, bias=False)
        lin(p + "self_attn.v_proj", d.llm_hkv * d.llm_hd, d.llm_h, bias=False)
        lin(p + "self_attn.o_proj", d.llm_h, d.llm_hq * d.llm_hd, bias=False)
        if d.qk_norm:
            ln(p + "self_attn.q_norm", d.llm_hd, bias=False)
            ln(p + "self_attn.k_norm", d.llm_hd, bias=False)
        lin(p + "mlp.gate_proj", d.llm_inter, d.llm_h, bias=False)
        lin(p + "mlp.up_proj", d.llm_inter, d.llm_h, bias=False)
        lin(p + "mlp.down_proj", d.llm_h, d.llm_inter, bias=False)
    ln(LLM + "model.norm", d.llm_h, bias=False)
    if not d.tie_embeddings:
        lin(LLM + "lm_head", d.vocab, d.llm_h, bias=False)
    # LoRA adapters.  peft initialises B = 0 (the adapter starts as the identity); a seeded non-zero B here so that parity
    # tests exercise the path (tests that need the peft init zero it themselves).
    if d.lora_r:
        for i in range(d.llm_layers):
            for m, out_f in (("q", d.llm_hq * d.llm_hd), ("k", d.llm_hkv * d.llm_hd), ("v", d.llm_hkv * d.llm_hd)):
                p = f"{LLM}model.layers.{i}.self_attn.{m}_proj."
                w[p + "lora_A.default.weight"] = (torch.rand(d.lora_r, d.llm_h, generator=g) * 2 - 1) / math.sqrt(d.llm_h)
                w[p + "lora_B.default.weight"] = 0.3 * (torch.rand(out_f, d.lora_r, generator=g) * 2 - 1) / math.sqrt(d.lora_r)
    return w


# ----------------------------------------------------------------------------- log-mel (A1)
def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    mels = 3.0 * f / 200.0
    logstep = 27.0 / np.log(6.4)
    hi = f >= 1000.0
    mels = np.where(hi, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * logstep, mels)
    return mels


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f = 200.0 * m / 3.0
    logstep = np.log(6.4) / 27.0
    hi = m >= 15.0
    return np.where(hi, 1000.0 * np.exp(logstep * (m - 15.0)), f)


def mel_filter_bank(n_mels: int, n_freq: int = 201, sr: int = 16000, fmax: float = 8000.0) -> np.ndarray:
    """Slaney-scale, slaney-normalised triangular filters, shape [n_freq, n_mels] float32
    (restates ``TF:audio_utils.py:638-729`` as configured by ``feature_extraction_whisper.py:94-103``)."""
    mel_pts = np.linspace(_hz_to_mel_slaney(0.0), _hz_to_mel_slaney(fmax), n_mels + 2)
    filt_f = _mel_to_hz_slaney(mel_pts)
    fft_f = np.linspace(0, sr // 2, n_freq)
    fdiff = np.diff(filt_f)
    slopes = filt_f[None, :] - fft_f[:, None]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (filt_f[2:n_mels + 2] - filt_f[:n_mels])
    fb = fb * enorm[None, :]
    return fb.astype(np.float32)


N_FFT, HOP, N_SAMPLES = 400, 160, 480000


def logmel(wave: Tensor, n_mels: int) -> Tensor:
    """[B, n] f32 waveform → [B, n_mels, 3000] f32 (zero-pad / truncate to 30 s first)."""
    wave = wave.float()
    B, n = wave.shape
    if n < N_SAMPLES:
        wave = F.pad(wave, (0, N_SAMPLES - n))
    wave = wave[:, :N_SAMPLES]
    window = torch.hann_window(N_FFT)
    stft = torch.stft(wave, N_FFT, HOP, window=window, return_complex=True)
    mag = stft[..., :-1].abs() ** 2                                    # drop last frame (H12)
    fb = torch.from_numpy(mel_filter_bank(n_mels))
    mel = fb.T @ mag
    log_spec = torch.clamp(mel, min=1e-10).log10()
    mx = log_spec.amax(dim=(1, 2), keepdim=True)                       # per-clip global max (H12)
    log_spec = torch.maximum(log_spec, mx - 8.0)
    return (log_spec + 4.0) / 4.0


# ----------------------------------------------------------------------------- Whisper (A3, A4)
def _mha(q, k, v, heads, scale_q: Optional[float] = None, scale_s: Optional[float] = None, mask=None):
    B, Sq, D = q.shape
    Sk = k.shape[1]
    hd = D // heads
    q = q.view(B, Sq, heads, hd).transpose(1, 2)
    k = k.view(B, Sk, heads, hd).transpose(1, 2)
    v = v.view(B, Sk, heads, hd).transpose(1, 2)
    if scale_q is not None:
        q = q * scale_q
    s = _mm(q, k.transpose(-1, -2))
    if scale_s is not None:
        s = s * scale_s
    if mask is not None:
        s = s + mask
    p = _softmax(s)
    o = _mm(p, v)
    return o.transpose(1, 2).reshape(B, Sq, D)


def whisper_stem(w, d: Dims, mel: Tensor) -> Tensor:
    if mel.shape[-1] != 2 * d.enc_T:
        raise ValueError(f"Whisper expects the mel input features to be of length {2 * d.enc_T}, "
                         f"but found {mel.shape[-1]}.")
    x = F.gelu(_conv1d(mel, w[ENC + "conv1.weight"], w[ENC + "conv1.bias"], padding=1))
    x = F.gelu(_conv1d(x, w[ENC + "conv2.weight"], w[ENC + "conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1)
    return x + w[ENC + "embed_positions.weight"][: d.enc_T]


def whisper_layer(w, d: Dims, i: int, x: Tensor) -> Tensor:
    p = f"{ENC}layers.{i}."
    hd = d.enc_d // d.enc_heads
    h = _ln(x, d.enc_d, w[p + "self_attn_layer_norm.weight"], w[p + "self_attn_layer_norm.bias"], 1e-5)
    q = _lin(h, w[p + "self_attn.q_proj.weight"], w[p + "self_attn.q_proj.bias"])
    k = _lin(h, w[p + "self_attn.k_proj.weight"])                      # no bias (H6)
    v = _lin(h, w[p + "self_attn.v_proj.weight"], w[p + "self_attn.v_proj.bias"])
    a = _mha(q, k, v, d.enc_heads, scale_q=hd ** -0.5)                 # q scaled before QK^T (H6)
    x = x + _lin(a, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"])
    h = _ln(x, d.enc_d, w[p + "final_layer_norm.weight"], w[p + "final_layer_norm.bias"], 1e-5)
    h = F.gelu(_lin(h, w[p + "fc1.weight"], w[p + "fc1.bias"]))
    return x + _lin(h, w[p + "fc2.weight"], w[p + "fc2.bias"])


def whisper_taps(w, d: Dims, mel: Tensor) -> List[Tensor]:
    """Hidden states after each tapped layer (no final layer_norm, H2)."""
    x = whisper_stem(w, d, mel)
    taps = []
    for i in range(d.enc_layers):
        x = whisper_layer(w, d, i, x)
        if i in d.taps:
            taps.append(x)
    return taps


# ----------------------------------------------------------------------------- Q-Former (A5, A6)
def _bert_attn_block(w, p, d: Dims, x: Tensor, kv: Tensor) -> Tensor:
    hd = d.enc_d // d.qf_heads
    q = _lin(x, w[p + "self.query.weight"], w[p + "self.query.bias"])
    k = _lin(kv, w[p + "self.key.weight"], w[p + "self.key.bias"])
    v = _lin(kv, w[p + "self.value.weight"], w[p + "self.value.bias"])
    a = _mha(q, k, v, d.qf_heads, scale_s=1.0 / math.sqrt(hd))        # bidirectional / unmasked (H5)
    o = _lin(a, w[p + "output.dense.weight"], w[p + "output.dense.bias"])
    return _ln(o + x, d.enc_d, w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], 1e-12)


def qformer_layer(w, d: Dims, i: int, x: Tensor, enc: Tensor) -> Tensor:
    p = f"{CON}qformer.layer.{i}."
    x = _bert_attn_block(w, p + "attention.", d, x, x)
    x = _bert_attn_block(w, p + "crossattention.", d, x, enc)
    h = F.gelu(_lin(x, w[p + "intermediate.dense.weight"], w[p + "intermediate.dense.bias"]))
    o = _lin(h, w[p + "output.dense.weight"], w[p + "output.dense.bias"])
    return _ln(o + x, d.enc_d, w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], 1e-12)


def qformer(w, d: Dims, j: int, enc: Tensor) -> Tensor:
    """Tap j: prompt j through all Q-Former layers (same weights for every tap)."""
    x = w[f"{CON}layer_prompts.{j}"].expand(enc.shape[0], -1, -1)
    for i in range(d.qf_layers):
        x = qformer_layer(w, d, i, x, enc)
    return x[:, : d.prompt_size]


def mix_proj(w, d: Dims, tap_outs: List[Tensor]) -> Tensor:
    x = torch.stack(tap_outs, dim=0).permute(1, 2, 0, 3)              # [B, K, taps, d]
    nw = torch.softmax(w[CON + "layer_weights"], dim=-1).unsqueeze(-1)
    x = (x * nw).sum(dim=2)
    x = _ln(x, d.enc_d, w[CON + "proj.0.weight"], w[CON + "proj.0.bias"], 1e-5)
    return _lin(x, w[CON + "proj.1.weight"], w[CON + "proj.1.bias"])


def perception(w, d: Dims, mel: Tensor, keep: Optional[dict] = None) -> Tensor:
    taps = whisper_taps(w, d, mel)
    outs = [qformer(w, d, j, t) for j, t in enumerate(taps)]
    af = mix_proj(w, d, outs)
    if keep is not None:
        keep["taps"], keep["qformer_out"], keep["audio_features"] = taps, outs, af
    return af


# ----------------------------------------------------------------------------- embed + splice (A7)
def embed_splice(w, d: Dims, input_ids: Tensor, audio_features: Optional[Tensor],
                 batch_transcription_ids: List[Tensor], batch_start_positions: List[Tuple[int, int]]) -> Tensor:
    emb = w[LLM + "model.embed_tokens.weight"]
    if _AC:
        emb = emb.to(BF16)                                             # the LLM is loaded in bf16 (modeling_desta25.py:715)
    x = F.embedding(input_ids, emb)
    if audio_features is None or len(batch_start_positions) == 0:
        return x
    assert len(batch_start_positions) == len(batch_transcription_ids) == audio_features.shape[0]
    out = x.clone()
    for a, (row, start) in enumerate(batch_start_positions):
        start = int(start)
        tr = F.embedding(batch_transcription_ids[a].reshape(-1), emb).detach()
        seg = torch.cat([audio_features[a], tr], dim=0)
        assert seg.shape[0] == d.prompt_size + tr.shape[0]
        idx = torch.arange(start, start + seg.shape[0])
        out = out.index_put((torch.tensor(int(row)), idx), seg.to(out.dtype))
    return out


# ----------------------------------------------------------------------------- LLM (A8)
def rope_inv_freq(d: Dims) -> Tensor:
    dim = d.llm_hd
    inv = 1.0 / (d.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.float64) / dim))
    if d.rope_llama3 is not None:
        factor, lo, hi, old = d.rope_llama3
        low_wl, high_wl = old / lo, old / hi
        wl = 2 * math.pi / inv
        inv_l = torch.where(wl > low_wl, inv / factor, inv)
        smooth = (old / wl - lo) / (hi - lo)
        sm = (1 - smooth) * inv_l / factor + smooth * inv_l
        med = ~(wl < high_wl) & ~(wl > low_wl)
        inv = torch.where(med, sm, inv_l)
    return inv.float()


def _rmsnorm(x, weight, eps):
    v = x.float().pow(2).mean(-1, keepdim=True)
    return weight.to(x.dtype) * (x.float() * torch.rsqrt(v + eps)).to(x.dtype)     # bf16 weight * bf16 under autocast_bf16


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def _lora_lin(w, d: Dims, name: str, h: Tensor, masks: Optional[dict]) -> Tensor:
    """A q/k/v projection with its optional LoRA adapter: ``base(x) + lora_B(lora_A(dropout(x))) * alpha / r``.
    ``masks[name]`` = keep mask [B,S,h] * 1/(1-p) of the adapter's OWN dropout module (training; None = eval / p = 0)."""
    y = _lin(h, w[name + ".weight"])
    a = w.get(name + ".lora_A.default.weight")
    if a is None:
        return y
    x = h                                                            # peft casts x to the adapter dtype (fp32); under autocast the linear casts back to bf16
    if masks is not None and masks.get(name) is not None:
        x = x * masks[name].to(x.dtype)
    return y + (_lin(_lin(x, a), w[name + ".lora_B.default.weight"]) * (d.lora_alpha / d.lora_r)).to(y.dtype)


def llm_forward(w, d: Dims, inputs_embeds: Tensor, attention_mask: Tensor, keep: Optional[dict] = None,
                position_ids: Optional[Tensor] = None, lora_masks: Optional[dict] = None, layer_hook=None) -> Tensor:
    """Returns logits [B,S,V].  position_ids = arange(S) for every row (H7) unless given ([B,S], the
    generate() path); additive causal mask AND left-pad key mask, as ``create_causal_mask`` builds it."""
    B, S, _ = inputs_embeds.shape
    x = inputs_embeds
    if position_ids is None:
        pos = torch.arange(S, dtype=torch.float32)
        fr = torch.outer(pos, rope_inv_freq(d))
        cos = torch.cat([fr, fr], -1).cos()[None, None]
        sin = torch.cat([fr, fr], -1).sin()[None, None]
    else:
        fr = position_ids.float()[:, :, None] * rope_inv_freq(d)[None, None, :]            # [B,S,hd/2]
        cos = torch.cat([fr, fr], -1).cos()[:, None]
        sin = torch.cat([fr, fr], -1).sin()[:, None]
    neg = torch.finfo(BF16 if _AC else torch.float32).min      # HF builds the mask with the model dtype's min
    causal = torch.ones(S, S, dtype=torch.bool).tril()
    allowed = causal[None, None] & attention_mask.bool()[:, None, None, :]
    mask = torch.zeros(B, 1, S, S).masked_fill(~allowed, neg)
    if _AC:                                   # HF: cos / sin cast to the activations' dtype (TF:…modeling_llama.py:117); mask in model dtype
        x, cos, sin, mask = x.to(BF16), cos.to(BF16), sin.to(BF16), mask.to(BF16)
    rep = d.llm_hq // d.llm_hkv
    for i in range(d.llm_layers):
        p = f"{LLM}model.layers.{i}."
        h = _rmsnorm(x, w[p + "input_layernorm.weight"], d.rms_eps)
        q = _lora_lin(w, d, p + "self_attn.q_proj", h, lora_masks).view(B, S, d.llm_hq, d.llm_hd)
        k = _lora_lin(w, d, p + "self_attn.k_proj", h, lora_masks).view(B, S, d.llm_hkv, d.llm_hd)
        v = _lora_lin(w, d, p + "self_attn.v_proj", h, lora_masks).view(B, S, d.llm_hkv, d.llm_hd)
        if d.qk_norm:
            q = _rmsnorm(q, w[p + "self_attn.q_norm.weight"], d.rms_eps)
            k = _rmsnorm(k, w[p + "self_attn.k_norm.weight"], d.rms_eps)
        q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        k = k.repeat_interleave(rep, dim=1)
        v = v.repeat_interleave(rep, dim=1)
        s = _mm(q, k.transpose(-1, -2)) * (d.llm_hd ** -0.5) + mask
        a = _mm(_softmax(s).to(q.dtype), v)   # eager_attention_forward: softmax(dtype=float32).to(query.dtype)
        a = a.transpose(1, 2).reshape(B, S, d.llm_hq * d.llm_hd)
        x = x + _lin(a, w[p + "self_attn.o_proj.weight"])
        h = _rmsnorm(x, w[p + "post_attention_layernorm.weight"], d.rms_eps)
        g = _lin(h, w[p + "mlp.gate_proj.weight"])
        u = _lin(h, w[p + "mlp.up_proj.weight"])
        x = x + _lin(F.silu(g) * u, w[p + "mlp.down_proj.weight"])
        if layer_hook is not None:            # ORCA deep injection wraps every decoder layer's forward (modeling_desta25.py:1101-1141)
            x = layer_hook(i, x)
        if keep is not None:
            keep.setdefault("llm_hidden", []).append(x)
    x = _rmsnorm(x, w[LLM + "model.norm.weight"], d.rms_eps)
    head = w[LLM + "model.embed_tokens.weight"] if d.tie_embeddings else w[LLM + "lm_head.weight"]
    return _lin(x, head)


def generation_position_ids(attention_mask: Tensor) -> Tensor:
    """``GenerationMixin._prepare_position_ids_for_generation`` (TF:generation/utils.py:751-773):
    cumsum(mask) - 1, pads at 0."""
    pos = attention_mask.long().cumsum(-1) - 1
    return pos.masked_fill(attention_mask == 0, 0)


def greedy_generate(w, d: Dims, inputs_embeds: Tensor, attention_mask: Tensor, max_new_tokens: int, pad_token_id: int,
                    eos_token_ids: Optional[List[int]] = None, forced_tokens: Optional[Tensor] = None, layer_hook=None):
    """Restatement of ``llm_model.generate(inputs_embeds=…, attention_mask=…, do_sample=False)``
    (reference call site modeling_desta25.py:1419-1427) WITHOUT a KV cache: every step re-runs the whole
    prefix, so the result is what any correct cache must reproduce.  Returns (new tokens [B,n], per-step
    logits [n,B,V]).  Finished sequences emit pad_token_id; stops once all sequences have emitted an EOS."""
    B = inputs_embeds.shape[0]
    emb = w[LLM + "model.embed_tokens.weight"]
    x, mask = inputs_embeds, attention_mask.long()
    unfinished = torch.ones(B, dtype=torch.long)
    toks, step_logits = [], []
    for t in range(max_new_tokens):
        logits = llm_forward(w, d, x, mask, position_ids=generation_position_ids(mask), layer_hook=layer_hook)[:, -1]
        step_logits.append(logits)
        nxt = logits.argmax(-1) if forced_tokens is None else forced_tokens[:, t]
        nxt = nxt * unfinished + pad_token_id * (1 - unfinished)
        toks.append(nxt)
        if eos_token_ids:
            unfinished = unfinished & ~torch.isin(nxt, torch.tensor(eos_token_ids)).long()
            if int(unfinished.max()) == 0:
                break
        x = torch.cat([x, emb[nxt][:, None].to(x.dtype)], 1)
        mask = torch.cat([mask, torch.ones(B, 1, dtype=torch.long)], 1)
    return torch.stack(toks, 1), torch.stack(step_logits)


def causal_lm_loss(logits: Tensor, labels: Tensor) -> Tensor:
    """``ForCausalLMLoss``: fp32 logits, labels shifted left by one (pad -100), token-mean (H8)."""
    logits = logits.float()
    labels = F.pad(labels, (0, 1), value=-100)[..., 1:].contiguous()
    return F.cross_entropy(logits.view(-1, logits.shape[-1]), labels.view(-1), ignore_index=-100, reduction="mean")


def model_forward(w, d: Dims, batch: dict, keep: Optional[dict] = None, lora_masks: Optional[dict] = None):
    """``DeSTA25AudioModel.forward`` (qformer_1 path): returns (loss|None, logits)."""
    ids, am = batch["input_ids"], batch["attention_mask"]
    feats = batch.get("batch_features")
    starts = batch.get("batch_start_positions", [])
    af = None
    if len(starts) > 0:
        af = perception(w, d, feats.float(), keep)
    x = embed_splice(w, d, ids, af, batch.get("batch_transcription_ids", []), starts)
    if keep is not None:
        keep["inputs_embeds"] = x
    logits = llm_forward(w, d, x, am, keep, lora_masks=lora_masks)
    loss = causal_lm_loss(logits, batch["labels"]) if batch.get("labels") is not None else None
    return loss, logits


# ----------------------------------------------------------------------------- optimiser (A11)
def decay_mask(names: List[str]) -> List[bool]:
    """Trainer's decay group: every parameter whose name has no 'bias' / 'LayerNorm' /
    'layernorm' / 'norm' style match and is not inside a LayerNorm module
    (``TF:trainer.py:1181-1195, 1305-1315``).  ``layer_prompts`` / ``layer_weights`` ARE decayed;
    ``proj.0`` is an nn.LayerNorm module so both of its tensors are not."""
    out = []
    for n in names:
        nd = (("bias" in n) or ("LayerNorm" in n) or (".proj.0." in n) or ("layernorm" in n) or ("layer_norm" in n)
              or (".global_proj.0." in n) or (".local_ln." in n) or n.endswith(".ln.weight"))          # ORCA: nn.LayerNorm modules under other names
        out.append(not nd)
    return out


def clip_grad_norm(grads: List[Tensor], max_norm: float = 1.0, f64: bool = False) -> Tensor:
    """``torch.nn.utils.clip_grad_norm_`` semantics (in place); returns the total norm.
    ``f64=True`` accumulates the norm in float64 (the fp32 CPU norm of multi-million-element tensors
    carries ~1e-4 relative summation error of its own, which would otherwise dominate a comparison)."""
    if f64:
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    else:
        total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g.float(), 2) for g in grads]), 2)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adafactor_init(params: List[Tensor]) -> List[dict]:
    st = []
    for p in params:
        if p.dim() >= 2:
            st.append({"step": 0, "row": torch.zeros(p.shape[:-1]), "col": torch.zeros(p.shape[:-2] + p.shape[-1:])})
        else:
            st.append({"step": 0, "sq": torch.zeros_like(p)})
    return st


def adafactor_step(params: List[Tensor], grads: List[Tensor], state: List[dict], lr: float,
                   wd: List[float], eps1: float = 1e-30, clip_threshold: float = 1.0,
                   decay_rate: float = -0.8, f64_stats: bool = False) -> None:
    """``transformers.optimization.Adafactor.step`` with scale_parameter=False,
    relative_step=False, beta1=None (``TF:optimization.py:1203-1294``).
    ``f64_stats=True`` accumulates the reductions (row/col means, RMS) in float64: torch's fp32 CPU
    reductions over multi-million-element tensors carry ~1e-4 relative error of their own."""
    for p, g, s, wdk in zip(params, grads, state, wd):
        g = g.float()
        s["step"] += 1
        beta2t = 1.0 - math.pow(s["step"], decay_rate)
        upd = g * g + eps1
        if p.dim() >= 2:
            rm = upd.double().mean(dim=-1).float() if f64_stats else upd.mean(dim=-1)
            cm = upd.double().mean(dim=-2).float() if f64_stats else upd.mean(dim=-2)
            s["row"].mul_(beta2t).add_(rm, alpha=1.0 - beta2t)
            s["col"].mul_(beta2t).add_(cm, alpha=1.0 - beta2t)
            rmean = s["row"].double().mean(dim=-1, keepdim=True).float() if f64_stats else s["row"].mean(dim=-1, keepdim=True)
            r = (s["row"] / rmean).rsqrt().unsqueeze(-1)
            c = s["col"].unsqueeze(-2).rsqrt()
            upd = (r * c) * g
        else:
            s["sq"].mul_(beta2t).add_(upd, alpha=1.0 - beta2t)
            upd = s["sq"].rsqrt() * g
        rms = (upd.double().norm(2).float() if f64_stats else upd.norm(2)) / math.sqrt(upd.numel())
        upd = upd / torch.clamp(rms / clip_threshold, min=1.0)
        upd = upd * lr
        if wdk != 0:
            p.add_(p, alpha=-wdk * lr)
        p.add_(-upd)


def linear_warmup_lr(step: int, base_lr: float, warmup: int, total: int) -> float:
    """``get_linear_schedule_with_warmup`` (``TF:optimization.py:101-104``); ``step`` = number of
    scheduler steps already taken."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    return base_lr * max(0.0, (total - step) / max(1, total - warmup))


def train_step(w: Dict[str, Tensor], d: Dims, batch: dict, opt_state: List[dict], lr: float,
               weight_decay: float = 0.01, max_grad_norm: float = 1.0, keep: Optional[dict] = None, autocast: bool = False):
    """One optimiser step in the HF-Trainer order (``TF:trainer.py:1722-1797``):
    forward → loss → backward → clip_grad_norm_(1.0) → Adafactor → (caller steps the schedule)."""
    names = trainable_names(d)
    params = [w[n] for n in names]
    for p in params:
        p.requires_grad_(True)
        p.grad = None
    if autocast:                          # HF Trainer(bf16=True): autocast around compute_loss only; optimizer in fp32
        with autocast_bf16():
            loss, logits = model_forward(w, d, batch, keep)
    else:
        loss, logits = model_forward(w, d, batch, keep)
    loss.backward()
    grads = [p.grad.detach().clone() for p in params]
    for p in params:
        p.requires_grad_(False)
        p.grad = None
    raw = [g.clone() for g in grads]
    gnorm = clip_grad_norm(grads, max_grad_norm)
    wd = [weight_decay if m else 0.0 for m in decay_mask(names)]
    with torch.no_grad():
        adafactor_step(params, grads, opt_state, lr, wd)
    return loss.detach(), logits.detach(), dict(zip(names, raw)), gnorm


# ----------------------------------------------------------------------------- synthetic batch (§8d)
def synthetic_batch(d: Dims, B: int, S_ctx: int, S_tgt: int, seed: int = 1234, pad: Optional[List[int]] = None,
                    mel: bool = True) -> dict:
    """Seeded synthetic batch in the collate layout (``simple_dataset.py:248-264``): context ‖
    prompt_size placeholders ‖ targets, optional LEFT padding per row, labels -100 off-target."""
    g = torch.Generator().manual_seed(seed)
    K = d.prompt_size
    S = S_ctx + K + S_tgt
    pad = pad or [0] * B
    S_tot = S + max(pad)
    ids = torch.randint(3, d.vocab, (B, S_tot), generator=g)
    am = torch.ones(B, S_tot, dtype=torch.long)
    labels = torch.full((B, S_tot), -100, dtype=torch.long)
    starts = []
    for b in range(B):
        # row b: [pad]*pad[b] ‖ context (S_ctx + max(pad) - pad[b] tokens) ‖ placeholders ‖ target
        am[b, : pad[b]] = 0
        ids[b, : pad[b]] = 0
        start = S_tot - S_tgt - K
        starts.append((b, start))
        labels[b, S_tot - S_tgt:] = ids[b, S_tot - S_tgt:]
    out = {"input_ids": ids, "attention_mask": am, "labels": labels,
           "batch_start_positions": starts,
           "batch_transcription_ids": [torch.zeros(1, 0, dtype=torch.long) for _ in range(B)]}
    if mel:
        out["batch_features"] = 0.5 * torch.randn(B, d.n_mels, 2 * d.enc_T, generator=g)
    return out
