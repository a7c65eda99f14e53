"""CPU oracle of the ORCA-hybrid variant — TEST INFRASTRUCTURE, never imported by the product.

Plain-PyTorch fp32 restatement, on top of `desta_oracle` (same weight-dict convention, the reference's state-dict names), of
  * `ORCAHybridConnector`            /root/reference/desta/models/modeling_desta25.py:208-357
      global branch: learnable queries x tapped Whisper states through a BertEncoder Q-Former (same block as `qformer_1`),
      softmax layer mix, LayerNorm + Linear;  local branch: softmax mix of the SAME tapped states, Linear d -> h, Conv1d(k, stride,
      pad k // 2) over time, LayerNorm;
  * `compute_rope_freqs` / `apply_rotary_pos_emb`   :22-95   (one rotation over the WHOLE hidden vector, fractional positions)
  * `ORCAGatedCrossAttention`        :359-490  (nn.MultiheadAttention over the audio tokens, LayerNorm, data-dependent sigmoid gate,
      per-layer alignment loss against the transcription span in training mode)
  * deep injection after every decoder layer   :1052-1143,  the ORCA branch of `forward`   :775-841,
    `_prepare_inputs_for_llm` with `orca_global_num_tokens` spliced tokens   :940-1050,  `compute_orca_losses`   :1159-1206
  * the trainer's total loss = lm_loss + sum(orca_losses)   /root/reference/desta/trainer/desta_trainer.py:56-92.

Pinned by tests/golden/ref_orca_tiny.safetensors: the reference's own classes run by tests/golden/make_golden_from_reference.py
on a tiny local-config model (tests/test_oracle_pin.py::test_orca_*)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

import desta_oracle as O
from desta_oracle import CON, LLM, Dims, _lin, _ln, _mha

Tensor = torch.Tensor
XA = "orca_cross_attns."


@dataclass
class OrcaDims:
    """The `orca_*` fields of the reference's DeSTA25Config (:645-692) with its defaults."""
    global_num_tokens: int = 4
    local_enabled: bool = True
    local_downsample: int = 4
    local_kernel_size: int = 5
    gate_init: float = 0.1
    audio_position_scale: float = 2.5
    global_cross_attn: bool = False
    deep_injection_enabled: bool = True
    ortho_diversity_weight: float = 0.01
    ortho_weight_qformer_local: float = 0.01
    align_weight_local: float = 0.05


def trainable_names(d: Dims, o: OrcaDims) -> List[str]:
    """`named_parameters()` order of the trainable tensors (llm_model is frozen; `perception` is registered before
    `orca_cross_attns`, :732, :1084; inside the connector: __init__ order :243-287)."""
    n = [f"{CON}global_queries.{j}" for j in range(len(d.taps))] + [CON + "global_layer_weights"]
    for i in range(d.qf_layers):
        p = f"{CON}global_qformer.layer.{i}."
        for blk in ("attention", "crossattention"):
            for lin in ("self.query", "self.key", "self.value", "output.dense", "output.LayerNorm"):
                n += [f"{p}{blk}.{lin}.weight", f"{p}{blk}.{lin}.bias"]
        for lin in ("intermediate.dense", "output.dense", "output.LayerNorm"):
            n += [f"{p}{lin}.weight", f"{p}{lin}.bias"]
    n += [CON + "global_proj.0.weight", CON + "global_proj.0.bias", CON + "global_proj.1.weight", CON + "global_proj.1.bias"]
    if o.local_enabled:
        n += [CON + "local_layer_weights", CON + "local_proj_in.weight", CON + "local_proj_in.bias", CON + "local_conv.weight",
              CON + "local_conv.bias", CON + "local_ln.weight", CON + "local_ln.bias"]
    if o.deep_injection_enabled:
        for l in range(d.llm_layers):
            p = f"{XA}{l}."
            n += [p + "cross_attn.in_proj_weight", p + "cross_attn.in_proj_bias", p + "cross_attn.out_proj.weight", p + "cross_attn.out_proj.bias",
                  p + "gate_proj.0.weight", p + "gate_proj.0.bias", p + "gate_proj.2.weight", p + "gate_proj.2.bias", p + "ln.weight", p + "ln.bias"]
    return n


def init_weights(d: Dims, o: OrcaDims, seed: int = 0) -> Dict[str, Tensor]:
    """Frozen Whisper + LLM from `desta_oracle.init_weights`; ORCA tensors seeded at the reference's shapes.  (Not the reference's
    init distribution — xavier for MultiheadAttention, zero gate weight — but random everywhere, so that every path carries signal.)"""
    w = {k: v for k, v in O.init_weights(d, seed=seed).items() if not k.startswith(CON)}
    g = torch.Generator().manual_seed(seed + 9001)
    h, de = d.llm_h, d.enc_d

    def lin(name, out_f, in_f):
        k = 1.0 / math.sqrt(in_f)
        w[name + ".weight"] = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * k
        w[name + ".bias"] = (torch.rand(out_f, generator=g) * 2 - 1) * k

    def ln(name, n):
        w[name + ".weight"] = 1.0 + 0.1 * torch.randn(n, generator=g)
        w[name + ".bias"] = 0.1 * torch.randn(n, generator=g)
    for j in range(len(d.taps)):
        w[f"{CON}global_queries.{j}"] = torch.randn(1, o.global_num_tokens, de, generator=g)
    w[CON + "global_layer_weights"] = 0.3 * torch.randn(o.global_num_tokens, len(d.taps), generator=g)
    for i in range(d.qf_layers):
        p = f"{CON}global_qformer.layer.{i}."
        for blk in ("attention", "crossattention"):
            for m in ("self.query", "self.key", "self.value", "output.dense"):
                lin(p + blk + "." + m, de, de)
            ln(p + blk + ".output.LayerNorm", de)
        lin(p + "intermediate.dense", d.qf_inter, de)
        lin(p + "output.dense", de, d.qf_inter)
        ln(p + "output.LayerNorm", de)
    ln(CON + "global_proj.0", de)
    lin(CON + "global_proj.1", h, de)
    if o.local_enabled:
        w[CON + "local_layer_weights"] = 0.3 * torch.randn(len(d.taps), generator=g)
        lin(CON + "local_proj_in", h, de)
        k = o.local_kernel_size
        w[CON + "local_conv.weight"] = torch.randn(h, h, k, generator=g) / math.sqrt(h * k)
        w[CON + "local_conv.bias"] = 0.1 * torch.randn(h, generator=g)
        ln(CON + "local_ln", h)
    if o.deep_injection_enabled:
        for l in range(d.llm_layers):
            p = f"{XA}{l}."
            w[p + "cross_attn.in_proj_weight"] = (torch.rand(3 * h, h, generator=g) * 2 - 1) / math.sqrt(h)
            w[p + "cross_attn.in_proj_bias"] = 0.1 * torch.randn(3 * h, generator=g)
            lin(p + "cross_attn.out_proj", h, h)
            lin(p + "gate_proj.0", h // 4, h)
            lin(p + "gate_proj.2", 1, h // 4)
            w[p + "gate_proj.2.bias"] = w[p + "gate_proj.2.bias"] + o.gate_init
            ln(p + "ln", h)
    return w


# ----------------------------------------------------------------------------- connector (:289-357)
def global_qformer_layer(w, d: Dims, i: int, x: Tensor, enc: Tensor) -> Tensor:
    p = f"{CON}global_qformer.layer.{i}."
    x = O._bert_attn_block(w, p + "attention.", d, x, x)
    x = O._bert_attn_block(w, p + "crossattention.", d, x, enc)
    hmid = F.gelu(_lin(x, w[p + "intermediate.dense.weight"], w[p + "intermediate.dense.bias"]))
    out = _lin(hmid, w[p + "output.dense.weight"], w[p + "output.dense.bias"])
    return _ln(out + x, d.enc_d, w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], 1e-12)


def connector(w, d: Dims, o: OrcaDims, taps: List[Tensor]) -> Tuple[Tensor, Optional[Tensor]]:
    """taps: the tapped encoder states [B, T, d] (target layers, in order) -> (global [B, Kg, h], local [B, T', h] | None)."""
    B = taps[0].shape[0]
    outs = []
    for j, enc in enumerate(taps):
        x = w[f"{CON}global_queries.{j}"].expand(B, -1, -1)
        for i in range(d.qf_layers):
            x = global_qformer_layer(w, d, i, x, enc)
        outs.append(x)
    g = torch.stack(outs, dim=0).permute(1, 2, 0, 3)                       # [B, K, L, D]
    g = (g * torch.softmax(w[CON + "global_layer_weights"], dim=-1).unsqueeze(-1)).sum(dim=2)
    g = _ln(g, d.enc_d, w[CON + "global_proj.0.weight"], w[CON + "global_proj.0.bias"], 1e-5)
    g = _lin(g, w[CON + "global_proj.1.weight"], w[CON + "global_proj.1.bias"])
    if not o.local_enabled:
        return g, None
    t = torch.stack(taps, dim=0).permute(1, 2, 0, 3)                      # [B, T, L, D]
    lw = torch.softmax(w[CON + "local_layer_weights"], dim=-1).unsqueeze(-1)
    fused = (t * lw).sum(dim=2)
    loc = _lin(fused, w[CON + "local_proj_in.weight"], w[CON + "local_proj_in.bias"]).transpose(1, 2)
    loc = O._conv1d(loc, w[CON + "local_conv.weight"], w[CON + "local_conv.bias"], stride=o.local_downsample, padding=o.local_kernel_size // 2)
    loc = loc.transpose(1, 2)
    return g, _ln(loc, d.llm_h, w[CON + "local_ln.weight"], w[CON + "local_ln.bias"], 1e-5)


# ----------------------------------------------------------------------------- gated cross-attention (:359-490)
def rope_whole_vector(x: Tensor, theta: float, scale: float) -> Tensor:
    """`compute_rope_freqs` + `apply_rotary_pos_emb` (:22-95) on [B, T, H]: positions t / scale, half = H / 2, pair (i, i + half)."""
    T, H = x.shape[1], x.shape[2]
    half = H // 2
    inv = 1.0 / (theta ** (torch.arange(half, dtype=torch.float) / half))
    fr = (torch.arange(T, dtype=torch.float) / scale).unsqueeze(-1) * inv.unsqueeze(0)
    cos, sin = fr.cos().unsqueeze(0).to(x.dtype), fr.sin().unsqueeze(0).to(x.dtype)
    x1, x2 = x[..., :half], x[..., half:]
    return torch.cat([x1 * cos - x2 * sin, x1 * sin + x2 * cos], dim=-1)


def gated_cross_attention(w, d: Dims, o: OrcaDims, l: int, hs: Tensor, audio: Optional[Tensor],
                          trans_positions: Optional[List[Tuple[int, int, int]]], training: bool):
    """-> (hidden_out, layer_align_loss | None)."""
    if audio is None or audio.shape[1] == 0:
        return hs, None
    p = f"{XA}{l}."
    H = d.llm_h
    a = rope_whole_vector(audio.to(hs.dtype), d.rope_theta, o.audio_position_scale)
    wi, bi = w[p + "cross_attn.in_proj_weight"], w[p + "cross_attn.in_proj_bias"]
    q = _lin(hs, wi[:H], bi[:H])
    k = _lin(a, wi[H:2 * H], bi[H:2 * H])
    v = _lin(a, wi[2 * H:], bi[2 * H:])
    hd = H // d.llm_hq
    att = _mha(q, k, v, d.llm_hq, scale_s=1.0 / math.sqrt(hd))
    cross = _lin(att, w[p + "cross_attn.out_proj.weight"], w[p + "cross_attn.out_proj.bias"])
    cross = _ln(cross, H, w[p + "ln.weight"], w[p + "ln.bias"], 1e-5)
    gate = torch.sigmoid(_lin(F.gelu(_lin(hs, w[p + "gate_proj.0.weight"], w[p + "gate_proj.0.bias"])),
                              w[p + "gate_proj.2.weight"], w[p + "gate_proj.2.bias"]))
    align = None
    if training:
        with torch.no_grad():
            ap = F.normalize(a.mean(dim=1), dim=-1)
        if trans_positions is not None and len(trans_positions) > 0:
            pooled = [hs[b, s:e].mean(dim=0) for b, s, e in trans_positions if s < e and e <= hs.shape[1]]
            if pooled:
                tp = F.normalize(torch.stack(pooled, dim=0), dim=-1)
                n = min(ap.shape[0], tp.shape[0])
                align = (1 - F.cosine_similarity(ap[:n], tp[:n], dim=-1)).mean()
        else:
            tp = F.normalize(hs.mean(dim=1), dim=-1)
            align = (1 - F.cosine_similarity(ap, tp, dim=-1)).mean()
    return hs + gate * cross, align


# ----------------------------------------------------------------------------- losses (:1159-1206)
def orca_losses(o: OrcaDims, g: Optional[Tensor], loc: Optional[Tensor], layer_align: List[Tensor]) -> Dict[str, Tensor]:
    out: Dict[str, Tensor] = {}
    if g is not None:
        gn = F.normalize(g, dim=-1)
        if O._AC:                                        # (autocast policy: bmm / einsum operands are cast to bf16)
            gn = gn.to(torch.bfloat16)
        gram = torch.einsum("bkh,bqh->bkq", gn, gn).float()
        out["L_ortho_diversity"] = o.ortho_diversity_weight * ((gram - torch.eye(gram.shape[-1])) ** 2).mean()
    if g is not None and loc is not None:
        gn, ln_ = F.normalize(g, dim=-1), F.normalize(loc, dim=-1)
        if ln_.shape[1] > 100:
            ln_ = ln_[:, torch.linspace(0, ln_.shape[1] - 1, 100, dtype=torch.long), :]
        if O._AC:
            gn, ln_ = gn.to(torch.bfloat16), ln_.to(torch.bfloat16)
        out["L_ortho_qformer_local"] = o.ortho_weight_qformer_local * (torch.einsum("bgh,blh->bgl", gn, ln_).float() ** 2).mean()
    if layer_align:
        out["L_align_layerwise"] = o.align_weight_local * torch.stack(layer_align).mean()
    return out


# ----------------------------------------------------------------------------- model forward (:775-841)
def model_forward(w, d: Dims, o: OrcaDims, batch: dict, training: bool = True, keep: Optional[dict] = None):
    """-> (lm_loss | None, logits, orca_losses dict).  One audio per text row, in row order (the reference's cross-attention takes
    audio row b for text row b: `query=hidden_states [B, S, H], key=audio_local [N_audio, T', H]`)."""
    ids, am = batch["input_ids"], batch["attention_mask"]
    starts = [(int(r), int(s)) for r, s in batch.get("batch_start_positions", [])]
    trs = batch.get("batch_transcription_ids", [])
    emb = w[LLM + "model.embed_tokens.weight"]
    x = F.embedding(ids, emb)
    g = loc = None
    positions: Optional[List[Tuple[int, int, int]]] = None
    if starts:
        taps = O.whisper_taps(w, d, batch["batch_features"].float())
        g, loc = connector(w, d, o, taps)
        if keep is not None:
            keep["taps"], keep["global_tokens"], keep["local_tokens"] = taps, g, loc
        x = x.clone()
        positions = []
        Kg = g.shape[1]
        for a, (row, start) in enumerate(starts):
            tr = F.embedding(trs[a].reshape(-1), emb).detach()
            seg = torch.cat([g[a], tr], dim=0)
            x = x.index_put((torch.tensor(row), torch.arange(start, start + seg.shape[0])), seg.to(x.dtype))
            positions.append((row, start + Kg, start + Kg + tr.shape[0]))
    audio = None
    if starts and o.deep_injection_enabled:
        if o.global_cross_attn:
            audio = torch.cat([g, loc], dim=1) if loc is not None else g
        else:
            audio = loc
    aligns: List[Tensor] = []

    def hook(l: int, hs: Tensor) -> Tensor:
        out, al = gated_cross_attention(w, d, o, l, hs, audio, positions, training)
        if al is not None:
            aligns.append(al)
        return out
    if keep is not None:
        keep["inputs_embeds"] = x
    logits = O.llm_forward(w, d, x, am, keep, layer_hook=hook if audio is not None else None)
    loss = O.causal_lm_loss(logits, batch["labels"]) if batch.get("labels") is not None else None
    return loss, logits, orca_losses(o, g, loc, aligns)


def generate(w, d: Dims, o: OrcaDims, inputs: dict, max_new_tokens: int, pad_token_id: int, eos_token_ids: Optional[List[int]] = None,
             forced_tokens: Optional[Tensor] = None):
    """The ORCA branch of `_generate_step` (modeling_desta25.py:1358-1436) without a KV cache: global tokens spliced into the CONTEXT
    part of the batch, the audio tokens injected behind every decoder layer at every step (eval mode: no alignment loss).
    -> (new tokens [B, n], per-step logits [n, B, V])."""
    ids, am = inputs["context_input_ids"], inputs["context_attention_mask"]
    starts = [(int(r), int(s)) for r, s in inputs["context_batch_start_positions"]]
    trs = inputs["batch_transcription_ids"]
    emb = w[LLM + "model.embed_tokens.weight"]
    x = F.embedding(ids, emb).clone()
    taps = O.whisper_taps(w, d, inputs["batch_features"].float())
    g, loc = connector(w, d, o, taps)
    for a, (row, start) in enumerate(starts):
        seg = torch.cat([g[a], F.embedding(trs[a].reshape(-1), emb)], dim=0)
        x = x.index_put((torch.tensor(row), torch.arange(start, start + seg.shape[0])), seg.to(x.dtype))
    audio = None
    if o.deep_injection_enabled:
        audio = (torch.cat([g, loc], dim=1) if loc is not None else g) if o.global_cross_attn else loc

    def hook(l: int, hs: Tensor) -> Tensor:
        return gated_cross_attention(w, d, o, l, hs, audio, None, False)[0]
    return O.greedy_generate(w, d, x, am, max_new_tokens, pad_token_id, eos_token_ids, forced_tokens, layer_hook=hook if audio is not None else None)


def total_loss(lm_loss: Tensor, losses: Dict[str, Tensor]) -> Tensor:
    """desta_trainer.py:56-92: every ORCA term is added to the LM loss."""
    t = lm_loss
    for v in losses.values():
        t = t + v
    return t
