"""Decode-side measurement of `_generate_step` at the full model size: prompt pass + KV-cached greedy
decode on one MI355X.  Reports prompt ms, ms per generated token (whole batch), tokens/s, and the weight-streaming
rate of the decode GEMMs (bytes of frozen LLM weights read per token / token time; HBM peak ~8 TB/s).

  python tools/decode_bench.py [--config desta25_llama31-8B_Qformer6L] [--batch 8] [--ctx 64] [--prompt-tail 16] [--new 64]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "desta2.5-audio_amd"))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="desta25_llama31-8B_Qformer6L")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--ctx", type=int, default=64)
    ap.add_argument("--prompt-tail", type=int, default=16, help="text tokens after the audio span")
    ap.add_argument("--new", type=int, default=64)
    ap.add_argument("--repeat", type=int, default=3)
    a = ap.parse_args()
    from desta.models.modeling_desta25 import DeSTA25AudioModel, DeSTA25Config
    from desta.synthetic import FULL_CONFIGS, RandomWeights, synthetic_inputs, synthetic_waveform
    from desta import _hip as H
    dev = torch.device("cuda:0")
    cfg = DeSTA25Config(**FULL_CONFIGS[a.config])
    model = DeSTA25AudioModel(cfg, weights=RandomWeights(cfg, dev, seed=0), device=dev).eval()
    B = a.batch
    t = synthetic_inputs(cfg, B, a.ctx, a.prompt_tail, dev, seed=5)
    mel = H.logmel(synthetic_waveform(B, dev, seed=6), cfg.encoder_config.num_mel_bins)
    inputs = {"context_input_ids": t["input_ids"], "context_attention_mask": t["attention_mask"],
              "context_batch_start_positions": t["batch_start_positions"], "batch_features": mel,
              "batch_transcription_ids": t["batch_transcription_ids"]}
    S = t["input_ids"].shape[1]
    res = []
    for new in (1, a.new):
        best = None
        for _ in range(a.repeat):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ids = model._generate_step(inputs, pad_token_id=0, max_new_tokens=new, do_sample=False, eos_token_id=[])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        assert ids.shape == (B, new)
        res.append(best)
    prompt_ms = res[0] * 1e3
    tok_ms = (res[1] - res[0]) * 1e3 / (a.new - 1)
    c = cfg.llm_config
    per_layer = (c.num_attention_heads + 2 * c.num_key_value_heads) * c.head_dim * c.hidden_size + c.num_attention_heads * c.head_dim * c.hidden_size \
        + 3 * c.hidden_size * c.intermediate_size
    wbytes = 2 * (c.num_hidden_layers * per_layer + c.vocab_size * c.hidden_size)
    print(json.dumps({"workload": f"{a.config} generate B={B} prompt={S} new={a.new}", "prompt_ms": round(prompt_ms, 2),
                      "ms_per_token_step": round(tok_ms, 3), "tokens_per_s": round(B / tok_ms * 1e3, 1),
                      "weight_bytes_per_step": wbytes, "weight_stream_GBps": round(wbytes / tok_ms / 1e6, 1),
                      "hbm_peak_GBps": 8000}))


if __name__ == "__main__":
    main()
