"""Shared test helpers: oracle Dims -> product config, golden batch loading."""
import os

import torch
from safetensors.torch import load_file

import desta_oracle as O


def cfg_from_dims(d: O.Dims, dropout: float = 0.0):
    from desta.models.modeling_desta25 import DeSTA25Config
    scaling = None
    if d.rope_llama3 is not None:
        f, lo, hi, old = d.rope_llama3
        scaling = {"rope_type": "llama3", "factor": f, "low_freq_factor": lo, "high_freq_factor": hi,
                   "original_max_position_embeddings": old}
    llm = dict(model_type="qwen3" if d.qk_norm else "llama", hidden_size=d.llm_h, num_hidden_layers=d.llm_layers,
               num_attention_heads=d.llm_hq, num_key_value_heads=d.llm_hkv, head_dim=d.llm_hd,
               intermediate_size=d.llm_inter, vocab_size=d.vocab, rms_norm_eps=d.rms_eps, rope_theta=d.rope_theta,
               rope_scaling=scaling, tie_word_embeddings=d.tie_embeddings)
    enc = dict(num_mel_bins=d.n_mels, d_model=d.enc_d, encoder_layers=d.enc_layers, encoder_attention_heads=d.enc_heads,
               encoder_ffn_dim=d.enc_ffn, max_source_positions=d.enc_T)
    return DeSTA25Config(llm_model_id="local-llm", encoder_model_id="local-whisper", llm_config=llm, encoder_config=enc,
                         qformer_num_hidden_layers=d.qf_layers, prompt_size=d.prompt_size,
                         qformer_intermediate_size=d.qf_inter, target_layer_ids=list(d.taps), qformer_dropout=dropout)


def golden_batch(golden_dir, name):
    g = load_file(os.path.join(golden_dir, f"ref_tiny_{name}.safetensors"))
    n = g["starts"].shape[0]
    batch = {"input_ids": g["input_ids"], "attention_mask": g["attention_mask"], "labels": g["labels"],
             "batch_features": g["batch_features"],
             "batch_start_positions": [(int(b), int(s)) for b, s in g["starts"].tolist()],
             "batch_transcription_ids": [torch.zeros(1, 0, dtype=torch.long) for _ in range(n)]}
    return g, batch


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
