"""Reference point for the flash-attention kernels: the same shapes through torch's scaled_dot_product_attention (its ROCm
flash / efficient backends), forward and forward + backward.  python tools/sdpa_compare.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "desta2.5-audio_amd"))
import torch
import torch.nn.functional as F


def t_us(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for name, (B, Hq, Hkv, Sq, Sk, D, causal) in {"llm S=640 D=128 causal GQA 32/8": (8, 32, 8, 640, 640, 128, True),
                                              "whisper S=1500 D=64": (8, 20, 20, 1500, 1500, 64, False),
                                              "qformer cross Sq=64 Sk=1500 D=64": (32, 20, 20, 64, 1500, 64, False)}.items():
    q = torch.randn(B, Hq, Sq, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    kw = dict(is_causal=causal, enable_gqa=(Hq != Hkv))
    try:
        with torch.no_grad():
            f = t_us(lambda: F.scaled_dot_product_attention(q, k, v, **kw))
        o = F.scaled_dot_product_attention(q, k, v, **kw)
        do = torch.randn_like(o)

        def fb():
            o = F.scaled_dot_product_attention(q, k, v, **kw)
            o.backward(do)
        fbt = t_us(fb)
        fl = 4.0 * B * Hq * Sq * Sk * D * (0.5 if causal else 1.0)
        print(f"{name:36s} torch SDPA fwd {f:7.1f} us ({fl / f / 1e6:5.0f} TF/s)   fwd+bwd {fbt:7.1f} us  -> bwd ~{fbt - f:7.1f} us ({2.5 * fl / max(fbt - f, 1e-3) / 1e6:5.0f} TF/s)", flush=True)
    except Exception as e:                                    # noqa: BLE001
        print(f"{name:36s} torch SDPA failed: {type(e).__name__}: {str(e)[:200]}", flush=True)
