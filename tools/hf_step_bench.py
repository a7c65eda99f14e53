"""Baseline on the SAME GPU: the training step of the headline config composed from the stock PyTorch-ROCm / transformers
modules the reference itself composes (WhisperEncoder fp32 frozen, BertEncoder Q-Former 6L + LayerNorm/Linear projector fp32
trainable, LlamaForCausalLM bf16 frozen, bf16 autocast, clip_grad_norm_ 1.0, transformers.Adafactor) at true shapes with random
weights — B = 8 clips of 3000 mel frames, S = 640 tokens (64 context + 64 audio + 512 targets), no padding.  The log-mel front end is
left out of the timed step (the reference computes it in CPU DataLoader workers).  Prints ms per step.

    python tools/hf_step_bench.py [--steps 10] [--warmup 3] [--attn sdpa|eager]

This is a measuring stick written for this repo (tools/ only; nothing of the product imports it)."""
import argparse
import time

import torch
import torch.nn as nn


def build(attn: str, dev):
    from transformers import BertConfig, LlamaConfig, LlamaForCausalLM, WhisperConfig
    from transformers.models.bert.modeling_bert import BertEncoder
    from transformers.models.whisper.modeling_whisper import WhisperEncoder
    with torch.device(dev):
        wcfg = WhisperConfig(num_mel_bins=128, d_model=1280, encoder_layers=32, encoder_attention_heads=20, encoder_ffn_dim=5120,
                             max_source_positions=1500, attn_implementation=attn)
        enc = WhisperEncoder(wcfg).float().eval().requires_grad_(False)
        lcfg = LlamaConfig(hidden_size=4096, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8, intermediate_size=14336,
                           vocab_size=128256, rms_norm_eps=1e-5, rope_theta=500000.0, max_position_embeddings=8192, attn_implementation=attn)
        llm = LlamaForCausalLM(lcfg).to(torch.bfloat16).requires_grad_(False)
        qcfg = BertConfig(hidden_size=1280, num_hidden_layers=6, num_attention_heads=20, add_cross_attention=True, is_decoder=True)
        qcfg._attn_implementation = "eager"
        con = nn.ModuleDict(dict(qformer=BertEncoder(qcfg), norm=nn.LayerNorm(1280), proj=nn.Linear(1280, 4096)))
        con.prompts = nn.Parameter(torch.randn(4, 1, 64, 1280))
        con.mixw = nn.Parameter(torch.zeros(64, 4))
    llm.train()
    con.train()                                            # Q-Former dropout 0.1 as in the reference's training mode
    return enc, con, llm


def step(enc, con, llm, opt, mel, ids, labels, starts, taps=(7, 15, 23, 31)):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        with torch.no_grad():
            hs = enc(mel, output_hidden_states=True).hidden_states          # embeddings + one per layer
        outs = []
        for j, t in enumerate(taps):
            state = hs[t + 1]
            q = con.prompts[j].expand(state.size(0), -1, -1)
            outs.append(con["qformer"](hidden_states=q, encoder_hidden_states=state).last_hidden_state)
        stack = torch.stack(outs, 0).permute(1, 2, 0, 3)                     # [B, 64, taps, d]
        mixed = (stack * torch.softmax(con.mixw, -1).unsqueeze(-1)).sum(2)
        af = con["proj"](con["norm"](mixed))                                 # [B, 64, 4096]
        emb = llm.get_input_embeddings()(ids).clone()
        for b, s in enumerate(starts):
            emb[b, s:s + 64] = af[b].to(emb.dtype)
        loss = llm(inputs_embeds=emb, attention_mask=torch.ones_like(ids), labels=labels).loss
    loss.backward()
    torch.nn.utils.clip_grad_norm_([p for p in con.parameters()], 1.0)
    opt.step()
    opt.zero_grad(set_to_none=True)
    return loss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--attn", default="sdpa")
    a = ap.parse_args()
    from transformers.optimization import Adafactor
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    t0 = time.time()
    enc, con, llm = build(a.attn, dev)
    opt = Adafactor([p for p in con.parameters()], lr=1e-4, scale_parameter=False, relative_step=False, warmup_init=False, weight_decay=0.01)
    print(f"built in {time.time() - t0:.1f}s, trainable {sum(p.numel() for p in con.parameters()) / 1e6:.1f} M, HBM {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
    B, S = 8, 640
    g = torch.Generator(device=dev).manual_seed(1)
    batches = []
    for i in range(2):
        mel = torch.randn(B, 128, 3000, device=dev, generator=g)
        ids = torch.randint(3, 128256, (B, S), device=dev, generator=g)
        labels = torch.full((B, S), -100, device=dev)
        labels[:, 128:] = ids[:, 128:]
        batches.append((mel, ids, labels, [64] * B))
    for i in range(a.warmup):
        step(enc, con, llm, opt, *batches[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(enc, con, llm, opt, *batches[i % 2])
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / a.steps
    print(f"HF / PyTorch-ROCm step ({a.attn} attention, bf16 autocast): {ms:.1f} ms per step = {1e3 / ms:.2f} steps/s, loss {float(loss):.3f}, "
          f"peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)


if __name__ == "__main__":
    main()
