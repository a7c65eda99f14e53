"""One attention shape, forward (and optionally backward) launched a few times: the target of `rocprofv3 --pmc ...` passes.
  python tools/attn_pmc_case.py llm|whisper [fwd|bwd|both] [position-major]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "desta2.5-audio_amd"))
import torch
from desta import _hip as hip

which = sys.argv[1] if len(sys.argv) > 1 else "llm"
what = sys.argv[2] if len(sys.argv) > 2 else "fwd"
smajor = len(sys.argv) > 3
B, Hq, Hkv, S, D, causal = {"llm": (8, 32, 8, 640, 128, True), "whisper": (8, 20, 20, 1500, 64, False)}[which]
g = torch.Generator(device="cuda").manual_seed(1)
wq, wkv = Hq * D, Hkv * D
w = wq + 2 * wkv
qkv = torch.randn(B * S, w, generator=g, device="cuda").to(torch.bfloat16)
o = torch.zeros(B * S, wq, dtype=torch.bfloat16, device="cuda")
do = torch.randn(B * S, wq, generator=g, device="cuda").to(torch.bfloat16)
lse = torch.zeros(B, Hq, S, device="cuda")
dqkv = torch.zeros_like(qkv)
kw = dict(q_rs=B * w, k_rs=B * w, v_rs=B * w, o_rs=B * wq, q_bs=w, k_bs=w, v_bs=w, o_bs=wq) if smajor else {}
d = hip.attn_desc(qkv, qkv, qkv, o, lse, batch=B, hq=Hq, hkv=Hkv, sq=S, sk=S, hd=D, scale=D ** -0.5, causal=causal,
                  kv_start=None, q_off=0, k_off=wq, v_off=wq + wkv, **kw)
bkw = dict(do_rs=B * wq, dq_rs=B * w, dk_rs=B * w, dv_rs=B * w, do_bs=wq, dq_bs=w, dk_bs=w, dv_bs=w) if smajor else {}
for _ in range(10):
    if what in ("fwd", "both"):
        hip.attention_fwd(d)
    if what in ("bwd", "both"):
        hip.attention_bwd(d, do, dqkv, dqkv, dqkv, dq_off=0, dk_off=wq, dv_off=wq + wkv, **bkw)
torch.cuda.synchronize()
