"""Time the flash-attention kernels on the two shapes of the training step (events around repeated launches).
  python tools/attn_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "desta2.5-audio_amd"))
import torch
from desta import _hip as hip


def t_us(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for name, (B, Hq, Hkv, S, D, causal) in {"llm S=640 D=128 causal GQA": (8, 32, 8, 640, 128, True),
                                         "whisper S=1500 D=64": (8, 20, 20, 1500, 64, False),
                                         "qformer self S=64 D=64": (32, 20, 20, 64, 64, False),
                                         "qformer cross Sq=64 Sk=1500 D=64": (32, 20, 20, (64, 1500), 64, False)}.items():
    g = torch.Generator(device="cuda").manual_seed(1)
    wq, wkv = Hq * D, Hkv * D
    if isinstance(S, tuple):                             # cross-attention: separate q [B*Sq, wq] and k|v [B*Sk, 2 wkv] buffers
        Sq, Sk = S
        q = torch.randn(B * Sq, wq, generator=g, device="cuda").to(torch.bfloat16)
        kv = torch.randn(B * Sk, 2 * wkv, generator=g, device="cuda").to(torch.bfloat16)
        o = torch.zeros(B * Sq, wq, dtype=torch.bfloat16, device="cuda")
        do = torch.randn(B * Sq, wq, generator=g, device="cuda").to(torch.bfloat16)
        lse = torch.zeros(B, Hq, Sq, device="cuda")
        dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
        for pd in (0.0, 0.1):                             # training runs the probability-dropout variants (BertConfig default 0.1)
            d = hip.attn_desc(q, kv, kv, o, lse, batch=B, hq=Hq, hkv=Hkv, sq=Sq, sk=Sk, hd=D, scale=D ** -0.5, q_off=0, k_off=0, v_off=wkv,
                              dropout_p=pd, dropout_seed=1234)
            f = t_us(lambda: hip.attention_fwd(d))
            bw = t_us(lambda: hip.attention_bwd(d, do, dq, dkv, dkv, dk_off=0, dv_off=wkv))
            fl = 4.0 * B * Hq * Sq * Sk * D
            print(f"{name + f' p={pd}':38s} fwd {f:7.1f} us ({fl / f / 1e6:6.0f} TF/s)   bwd {bw:7.1f} us ({2.5 * fl / bw / 1e6:6.0f} TF/s)", flush=True)
        continue
    qkv = torch.randn(B * S, wq + 2 * wkv, generator=g, device="cuda").to(torch.bfloat16)
    o = torch.zeros(B * S, wq, dtype=torch.bfloat16, device="cuda")
    do = torch.randn(B * S, wq, generator=g, device="cuda").to(torch.bfloat16)
    lse = torch.zeros(B, Hq, S, device="cuda")
    dqkv = torch.zeros_like(qkv)
    d = hip.attn_desc(qkv, qkv, qkv, o, lse, batch=B, hq=Hq, hkv=Hkv, sq=S, sk=S, hd=D, scale=D ** -0.5, causal=causal,
                      kv_start=None, q_off=0, k_off=wq, v_off=wq + wkv)
    f = t_us(lambda: hip.attention_fwd(d))
    hip.attention_set_option(0, 0)
    f4 = t_us(lambda: hip.attention_fwd(d))
    hip.attention_set_option(0, 1)
    extra = f"  [4-wave fwd {f4:6.1f} us"
    hip.attention_set_option(2, 1)
    extra += f"; 8-wave with the half-tile stagger {t_us(lambda: hip.attention_fwd(d)):6.1f} us"
    hip.attention_set_option(2, 0)
    if D == 64:
        hip.attention_set_option(1, 1)
        extra += f"; 8-wave at 2 blocks/CU {t_us(lambda: hip.attention_fwd(d)):6.1f} us"
        hip.attention_set_option(1, 0)
    if causal:
        # the training layout of the LLM: position-major token grid (row = s * B + b): row stride B * width, batch stride width
        d2 = hip.attn_desc(qkv, qkv, qkv, o, lse, batch=B, hq=Hq, hkv=Hkv, sq=S, sk=S, hd=D, scale=D ** -0.5, causal=causal,
                           kv_start=None, q_off=0, k_off=wq, v_off=wq + wkv, q_rs=B * (wq + 2 * wkv), k_rs=B * (wq + 2 * wkv),
                           v_rs=B * (wq + 2 * wkv), o_rs=B * wq, q_bs=wq + 2 * wkv, k_bs=wq + 2 * wkv, v_bs=wq + 2 * wkv, o_bs=wq)
        extra += f"; position-major layout: 8-wave {t_us(lambda: hip.attention_fwd(d2)):6.1f} us"
        hip.attention_set_option(0, 0)
        extra += f", 4-wave {t_us(lambda: hip.attention_fwd(d2)):6.1f} us"
        hip.attention_set_option(0, 1)
    extra += "]"
    bw = t_us(lambda: hip.attention_bwd(d, do, dqkv, dqkv, dqkv, dq_off=0, dk_off=wq, dv_off=wq + wkv))
    if causal and D == 128:
        hip.attention_set_option(5, 1)                                       # 64-key dK / dV blocks of two waves (default off)
        other = t_us(lambda: hip.attention_bwd(d, do, dqkv, dqkv, dqkv, dq_off=0, dk_off=wq, dv_off=wq + wkv))
        hip.attention_set_option(5, 0)
        print(f"   backward with option 5 (64-key dK / dV blocks): {other:7.1f} us (default above: {bw:7.1f} us)", flush=True)
    hip.attention_set_concurrent_bwd(False)
    bw_serial = t_us(lambda: hip.attention_bwd(d, do, dqkv, dqkv, dqkv, dq_off=0, dk_off=wq, dv_off=wq + wkv))
    hip.attention_set_concurrent_bwd(True)
    fl = 4.0 * B * Hq * S * S * D * (0.5 if causal else 1.0)
    print(f"{name:30s} fwd {f:7.1f} us ({fl / f / 1e6:6.0f} TF/s)   bwd {bw:7.1f} us ({2.5 * fl / bw / 1e6:6.0f} TF/s; dQ after dK/dV on one stream: {bw_serial:7.1f} us){extra}", flush=True)
