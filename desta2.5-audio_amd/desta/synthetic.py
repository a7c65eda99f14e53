"""Seeded synthetic weights / batches at TRUE shapes, generated directly in HBM.

There are no checkpoints or datasets offline: benchmarks and full-size property tests use
random-init weights of the named architecture and synthetic 30 s clips + token batches in the collate
layout (simple_dataset.py:248-264)."""
from __future__ import annotations

import math
import re
from typing import Dict

import torch

from .models.modeling_desta25 import CON, ENC, LLM, DeSTA25Config, connector_param_shapes


class RandomWeights:
    """Lazy `{name: tensor}` mapping: every tensor is generated on `device` when first asked for and not
    kept, so building an 8 B-parameter model never holds a second copy (frozen weights in bf16)."""

    def __init__(self, cfg: DeSTA25Config, device, seed: int = 0, with_connector: bool = False):
        self.cfg, self.device, self.seed = cfg, torch.device(device), seed
        self._connector = connector_param_shapes(cfg) if with_connector else {}
        self._gen = torch.Generator(device=self.device)

    def __contains__(self, name: str) -> bool:
        return name in self._connector

    def _shape(self, name: str):
        e, c = self.cfg.encoder_config, self.cfg.llm_config
        d, f = e.d_model, e.encoder_ffn_dim
        if name.startswith(ENC):
            n = name[len(ENC):]
            if n == "conv1.weight": return (d, e.num_mel_bins, 3)
            if n == "conv2.weight": return (d, d, 3)
            if n in ("conv1.bias", "conv2.bias"): return (d,)
            if n == "embed_positions.weight": return (e.max_source_positions, d)
            if "fc1.weight" in n: return (f, d)
            if "fc1.bias" in n: return (f,)
            if "fc2.weight" in n: return (d, f)
            if n.endswith("proj.weight"): return (d, d)
            return (d,)
        if name.startswith(LLM):
            n = name[len(LLM):]
            h, I, hd = c.hidden_size, c.intermediate_size, c.head_dim
            if n in ("model.embed_tokens.weight", "lm_head.weight"): return (c.vocab_size, h)
            if "q_proj" in n: return (c.num_attention_heads * hd, h)
            if "k_proj" in n or "v_proj" in n: return (c.num_key_value_heads * hd, h)
            if "o_proj" in n: return (h, c.num_attention_heads * hd)
            if "gate_proj" in n or "up_proj" in n: return (I, h)
            if "down_proj" in n: return (h, I)
            if "q_norm" in n or "k_norm" in n: return (hd,)
            return (h,)
        return self._connector[name]

    def __getitem__(self, name: str) -> torch.Tensor:
        shape = self._shape(name)
        g = self._gen
        g.manual_seed((self.seed * 1000003 + (hash_name(name) % 1000003)) & 0x7FFFFFFF)
        is_norm = ("layer_norm" in name or "layernorm" in name or name.endswith("norm.weight") or "LayerNorm" in name or ".proj.0." in name)
        if is_norm:
            if name.endswith("bias"):
                return 0.02 * torch.randn(shape, generator=g, device=self.device)
            return 1.0 + 0.02 * torch.randn(shape, generator=g, device=self.device)
        if name.endswith("bias"):
            return 0.02 * torch.randn(shape, generator=g, device=self.device)
        if "embed_positions" in name or "embed_tokens" in name:
            return (0.02 * torch.randn(shape, generator=g, device=self.device, dtype=torch.float32)).to(torch.bfloat16)
        if "layer_prompts" in name:
            return torch.randn(shape, generator=g, device=self.device)
        if "layer_weights" in name:
            return torch.zeros(shape, device=self.device)
        fan_in = int(math.prod(shape[1:]))
        w = torch.empty(shape, device=self.device, dtype=torch.bfloat16 if not name.startswith(CON) else torch.float32)
        w.uniform_(-1.0, 1.0, generator=g)
        return w * (1.0 / math.sqrt(fan_in))


def hash_name(name: str) -> int:
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def synthetic_inputs(cfg: DeSTA25Config, B: int, S_ctx: int, S_tgt: int, device, seed: int = 1234, S_tr: int = 0) -> Dict:
    """Token side of a collated batch on `device`: context ‖ audio placeholders (‖ S_tr transcription slots) ‖ targets, no
    padding, labels -100 in front of the targets.  S_tr > 0: every clip carries S_tr transcription token ids, spliced
    behind its audio tokens like the reference's collate lays them out (simple_dataset.py:248-264)."""
    g = torch.Generator(device=device).manual_seed(seed)
    K, V = cfg.audio_tokens + S_tr, cfg.llm_config.vocab_size
    S = S_ctx + K + S_tgt
    ids = torch.randint(3, V, (B, S), generator=g, device=device)
    labels = torch.full((B, S), -100, dtype=torch.long, device=device)
    labels[:, S_ctx + K:] = ids[:, S_ctx + K:]
    return {"input_ids": ids, "attention_mask": torch.ones(B, S, dtype=torch.long, device=device), "labels": labels,
            "batch_start_positions": [(b, S_ctx) for b in range(B)],
            "batch_transcription_ids": [ids[b:b + 1, S_ctx + cfg.audio_tokens:S_ctx + K].clone() for b in range(B)]}


def synthetic_waveform(B: int, device, seed: int = 1234, n: int = 480000) -> torch.Tensor:
    g = torch.Generator(device=device).manual_seed(seed)
    return (0.1 * torch.randn(B, n, generator=g, device=device)).clamp_(-1.0, 1.0)


_ENC_LARGE_V3 = dict(num_mel_bins=128, d_model=1280, encoder_layers=32, encoder_attention_heads=20,
                     encoder_ffn_dim=5120, max_source_positions=1500)      # large-v3 and large-v3-turbo share the ENCODER
_LLAMA31_8B = dict(model_type="llama", hidden_size=4096, num_hidden_layers=32, num_attention_heads=32,
                   num_key_value_heads=8, head_dim=128, intermediate_size=14336, vocab_size=128256,
                   rms_norm_eps=1e-5, rope_theta=500000.0,
                   rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0,
                                 "high_freq_factor": 4.0, "original_max_position_embeddings": 8192})


def _qwen3(h, layers, heads, kv, inter, tied, theta=1000000.0):
    return dict(model_type="qwen3", hidden_size=h, num_hidden_layers=layers, num_attention_heads=heads, num_key_value_heads=kv,
                head_dim=128, intermediate_size=inter, vocab_size=151936, rms_norm_eps=1e-6, rope_theta=theta,
                tie_word_embeddings=tied)


FULL_CONFIGS = {
    # BASELINE.json configs[1]/[2]: Whisper-large-v3 + Llama-3.1-8B, Q-Former 6L (SURVEY Appendix B dims)
    "desta25_llama31-8B_Qformer6L": dict(
        llm_model_id="DeSTA-ntu/Llama-3.1-8B-Instruct", encoder_model_id="openai/whisper-large-v3",
        qformer_num_hidden_layers=6, prompt_size=64, llm_config=_LLAMA31_8B, encoder_config=_ENC_LARGE_V3),
    # configs[3]: the large-v3-turbo encoder variant (turbo prunes the DECODER; the reference taps layers 7/15/23/31 of a
    # 32-layer encoder for both ids, modeling_desta25.py:140-143) -> the same kernels and tilings as configs[1]
    "desta25_llama31-8B_turbo_Qformer6L": dict(
        llm_model_id="DeSTA-ntu/Llama-3.1-8B-Instruct", encoder_model_id="openai/whisper-large-v3-turbo",
        qformer_num_hidden_layers=6, prompt_size=64, llm_config=_LLAMA31_8B, encoder_config=_ENC_LARGE_V3),
    # configs[4]: Qwen3-8B backbone
    "desta25_qwen3-8B_Qformer6L": dict(
        llm_model_id="Qwen/Qwen3-8B", encoder_model_id="openai/whisper-large-v3",
        qformer_num_hidden_layers=6, prompt_size=64, placeholder_token="<|video_pad|>",
        llm_config=_qwen3(4096, 36, 32, 8, 12288, False), encoder_config=_ENC_LARGE_V3),
    # the two Qwen3 Q-Former configs the reference ships (examples/train/config/desta25_qwen3-{4B,0.6b}_Qformer6L.yaml):
    # tied lm_head, hidden != heads x head_dim, turbo encoder id (Qwen3-4B-Instruct-2507: rope_theta 5e6 per its model card)
    "desta25_qwen3-4B_Qformer6L": dict(
        llm_model_id="Qwen/Qwen3-4B-Instruct-2507", encoder_model_id="openai/whisper-large-v3-turbo",
        qformer_num_hidden_layers=6, prompt_size=64, placeholder_token="<|video_pad|>",
        llm_config=_qwen3(2560, 36, 32, 8, 9728, True, theta=5000000.0), encoder_config=_ENC_LARGE_V3),
    "desta25_qwen3-0.6b_Qformer6L": dict(
        llm_model_id="Qwen/Qwen3-0.6B", encoder_model_id="openai/whisper-large-v3-turbo",
        qformer_num_hidden_layers=6, prompt_size=64, placeholder_token="<|video_pad|>",
        llm_config=_qwen3(1024, 28, 16, 8, 3072, True), encoder_config=_ENC_LARGE_V3),
}
# the three ORCA-hybrid configs the reference ships (examples/train/config/desta25_{llama31-8B,qwen3-0.6b,qwen3-4b}_ORCAHybrid.yaml):
# global + local tokens in the gated cross-attention behind every decoder layer, 0.05 loss weights, whisper-large-v3
_ORCA = dict(connector_mode="orca_hybrid", orca_enabled=True, orca_local_enabled=True, orca_global_cross_attn=True,
             orca_deep_injection_enabled=True, orca_local_downsample=4, orca_local_kernel_size=5, orca_gate_init=0.1,
             orca_audio_position_scale=2.5, orca_ortho_weight_global=0.05, orca_ortho_diversity_weight=0.05,
             orca_ortho_weight_qformer_local=0.05, orca_align_weight_local=0.05)
FULL_CONFIGS.update({
    "desta25_llama31-8B_ORCAHybrid": dict(
        llm_model_id="DeSTA-ntu/Llama-3.1-8B-Instruct", encoder_model_id="openai/whisper-large-v3", qformer_num_hidden_layers=6,
        prompt_size=64, orca_global_num_tokens=8, llm_config=_LLAMA31_8B, encoder_config=_ENC_LARGE_V3, **_ORCA),
    "desta25_qwen3-0.6b_ORCAHybrid": dict(
        llm_model_id="Qwen/Qwen3-0.6B", encoder_model_id="openai/whisper-large-v3", qformer_num_hidden_layers=6, prompt_size=64,
        placeholder_token="<|video_pad|>", orca_global_num_tokens=64, llm_config=_qwen3(1024, 28, 16, 8, 3072, True),
        encoder_config=_ENC_LARGE_V3, **_ORCA),
    # hidden 2560 / 32 heads: cross-attention head size 80, run zero-padded to 128 (OrcaHIP)
    "desta25_qwen3-4b_ORCAHybrid": dict(
        llm_model_id="Qwen/Qwen3-4B", encoder_model_id="openai/whisper-large-v3", qformer_num_hidden_layers=6, prompt_size=64,
        placeholder_token="<|video_pad|>", orca_global_num_tokens=64, llm_config=_qwen3(2560, 36, 32, 8, 9728, True),
        encoder_config=_ENC_LARGE_V3, **_ORCA),
})


# ------------------------------------------------------------------------------------------------ real-data-path stand-ins (bench.py --data wav)
class WordTokenizer:
    """Deterministic word-level tokenizer with the slice of the HF tokenizer protocol that `BaseCollateFn` / `BaseAudioTextDataset`
    touch (left padding, right truncation, `return_length`, special tokens kept whole): stands in for the LLM's tokenizer where no
    tokenizer files exist offline (`bench.py --data wav`; the tests use the same class shape, tests/helpers.py)."""
    padding_side = "left"
    eos_token = "<|eos|>"
    pad_token_id, eos_token_id = 0, 2
    pad_token = None
    _SPLIT = re.compile(r"<\|[^|\s]*\|>|<[a-z_]+>|[A-Za-z0-9']+|[^\sA-Za-z0-9]")

    class _Enc(dict):
        def to(self, device):
            return self

    def __init__(self, vocab_size: int = 128256):
        self.vocab_size = vocab_size
        self._fixed = {"<|pad|>": 0, "<|bos|>": 1, "<|eos|>": 2, "<|AUDIO|>": 3, "<|video_pad|>": 4, "<|start|>": 5, "<|end|>": 6,
                       "<|reserved_special_token_87|>": 7}

    def add_tokens(self, toks):
        return 0

    def tokenize(self, text, add_special_tokens=False, **kw):
        return self._SPLIT.findall(text)

    def convert_tokens_to_string(self, tokens):
        return " ".join(tokens)

    def convert_tokens_to_ids(self, tokens):
        import zlib
        one = isinstance(tokens, str)
        ids = [self._fixed.get(t, 8 + zlib.crc32(t.encode()) % (self.vocab_size - 8)) for t in ([tokens] if one else tokens)]
        return ids[0] if one else ids

    def encode(self, text, add_special_tokens=False, return_tensors=None, **kw):
        ids = self.convert_tokens_to_ids(self.tokenize(text))
        return torch.tensor([ids], dtype=torch.long).reshape(1, len(ids)) if return_tensors == "pt" else ids

    def __call__(self, texts, truncation=False, padding=False, max_length=None, return_tensors=None, return_length=False,
                 add_special_tokens=False, **kw):
        rows = [self.encode(t) for t in ([texts] if isinstance(texts, str) else texts)]
        if truncation and max_length is not None:
            rows = [r[:max_length] for r in rows]
        L = max(len(r) for r in rows)
        ids = torch.full((len(rows), L), self.pad_token_id, dtype=torch.long)
        am = torch.zeros(len(rows), L, dtype=torch.long)
        for i, r in enumerate(rows):
            if r:
                ids[i, L - len(r):] = torch.tensor(r)
                am[i, L - len(r):] = 1
        out = self._Enc({"input_ids": ids, "attention_mask": am})
        if return_length:
            out["length"] = torch.full((len(rows),), L, dtype=torch.long)
        return out

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=True, **kw):
        if messages and isinstance(messages[0], list):
            return [self.apply_chat_template(m, tokenize, add_generation_prompt) for m in messages]
        s = "".join(f"<|start|>{m['role']}\n{m['content']}<|end|>\n" for m in messages)
        return s + ("<|start|>assistant\n" if add_generation_prompt else "")

    def batch_decode(self, ids, skip_special_tokens=False):
        return [" ".join(str(int(t)) for t in row if not (skip_special_tokens and int(t) < 8)) for row in ids]


def write_synthetic_wav_dataset(root: str, n_files: int = 16, seed: int = 1234):
    """`n_files` 30-s RIFF/WAVE clips of seeded noise under `root` — even files 16 kHz mono PCM16, odd files 22.05 kHz STEREO PCM16
    (decode + channel average + polyphase resample on the host) — and the matching manifest records (id / prompt / response)."""
    import os
    import wave
    import numpy as np
    os.makedirs(root, exist_ok=True)
    rng = np.random.default_rng(seed)
    words = ["alpha", "bravo", "charlie", "delta", "echo", "foxtrot", "golf", "hotel", "india", "juliet", "kilo", "lima", "mike", "november"]
    records = []
    for i in range(n_files):
        sr, ch = (16000, 1) if i % 2 == 0 else (22050, 2)
        x = (0.1 * rng.standard_normal((30 * sr, ch))).clip(-1, 1)
        path = os.path.join(root, f"clip{i:03d}.wav")
        with wave.open(path, "wb") as w:
            w.setnchannels(ch)
            w.setsampwidth(2)
            w.setframerate(sr)
            w.writeframes((x * 32767).astype("<i2").tobytes())
        records.append(dict(id=f"clip{i:03d}.wav", prompt=" ".join(words[(i + j) % len(words)] for j in range(8)),
                            response=" ".join(words[(3 * i + j) % len(words)] for j in range(16))))
    return records
