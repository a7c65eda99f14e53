"""The connector (Q-Former 6L + tap mix + projector) alone at the headline shape: forward / backward wall time per call (HIP events)
and the host time to issue the launches.  `python tools/connector_bench.py [flags]`, flags = bit mask of what is ON:
1 one-pass cross-attention backward (desta_attention_set_option(4, .)), 2 d(K|V) written transposed + bias sums inside it,
4 K | V projections on a second stream; default 7 (the product's defaults).  Numbers: DESIGN.md §3 "The connector as a whole",
profiles/r03_attn_q64_onepass.log."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "desta2.5-audio_amd"))
from desta import _hip as H                                                                  # noqa: E402
from desta.models.modeling_desta25 import DeSTA25AudioModel, DeSTA25Config                  # noqa: E402
from desta.synthetic import FULL_CONFIGS, RandomWeights                                      # noqa: E402


def main():
    flags = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    dev = torch.device("cuda:0")
    cfg = DeSTA25Config(**FULL_CONFIGS["desta25_llama31-8B_Qformer6L"])
    cfg.llm_config.num_hidden_layers = 1                      # only the connector is of interest
    model = DeSTA25AudioModel(cfg, weights=RandomWeights(cfg, dev, seed=0), device=dev)
    con = model.connector
    B, e = 8, cfg.encoder_config
    enc = torch.randn(len(cfg.target_layer_ids), B * e.max_source_positions, e.d_model, device=dev).to(torch.bfloat16)
    con.refresh_weights()
    con.p_drop, con.seed_base = cfg.qformer_dropout, 12345
    d_af = torch.randn(B * cfg.prompt_size, cfg.llm_config.hidden_size, device=dev).to(torch.bfloat16)
    H.attention_set_option(4, flags & 1)
    con.xattn_transposed = bool(flags & 2)
    con.kv_side = bool(flags & 4)

    def t(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    print(f"flags {flags}: one-pass {flags & 1}, transposed d(K|V) {bool(flags & 2)}, K|V projections on a second stream {bool(flags & 4)}")
    print(f"connector forward  {t(lambda: con.forward(enc, B)):.3f} ms")
    print(f"connector backward {t(lambda: con.backward(d_af)):.3f} ms")
    torch.cuda.synchronize()
    t0 = time.time()
    con.forward(enc, B)
    t1 = time.time()
    con.backward(d_af)
    t2 = time.time()
    torch.cuda.synchronize()
    print(f"host issue time forward / backward {1e3 * (t1 - t0):.2f} / {1e3 * (t2 - t1):.2f} ms")


if __name__ == "__main__":
    main()
