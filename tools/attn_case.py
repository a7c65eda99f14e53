"""Diagnose one attention case: python tools/attn_case.py <case-index> <dq|full>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("desta2.5-audio_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from desta import _hip as hip
from test_gpu_ops import ATTN_CASES, _attn_ref, bf, rel_err

case = ATTN_CASES[int(sys.argv[1])]
mode = sys.argv[2]
B, Hq, Hkv, Sq, Sk, D, causal, pad = case
g = torch.Generator().manual_seed(Sq * 3 + Sk + D)
wq, wkv = Hq * D, Hkv * D
qb = bf(torch.randn(B * Sq, wq, generator=g))
kvb = bf(torch.randn(B * Sk, 2 * wkv, generator=g))
q = qb.float().reshape(B, Sq, Hq, D).clone().requires_grad_(True)
k = kvb[:, :wkv].float().reshape(B, Sk, Hkv, D).clone().requires_grad_(True)
v = kvb[:, wkv:].float().reshape(B, Sk, Hkv, D).clone().requires_grad_(True)
kvs = torch.tensor(pad, dtype=torch.int32) if pad is not None else None
ref = _attn_ref(q, k, v, D ** -0.5, causal, kvs)
do = bf(torch.randn(B * Sq, wq, generator=g))
ref.backward(do.float().view(B, Sq, Hq, D))
# generous guard padding around every device buffer so a small overrun shows up as a changed guard, not a fault
def dev(t, guard=1 << 20):
    buf = torch.full((guard * 2 + t.numel(),), 7.0, dtype=t.dtype, device="cuda")
    view = buf[guard:guard + t.numel()].view(t.shape)
    view.copy_(t)
    return buf, view
qbuf, qd = dev(qb); kvbuf, kvd = dev(kvb); dobuf, dod = dev(do)
obuf, o = dev(torch.zeros(B * Sq, wq, dtype=torch.bfloat16)); lbuf, lse = dev(torch.zeros(B, Hq, Sq))
dqbuf, dq = dev(torch.zeros(B * Sq, wq, dtype=torch.bfloat16)); dkvbuf, dkv = dev(torch.zeros(B * Sk, 2 * wkv, dtype=torch.bfloat16))
kvsd = kvs.cuda() if kvs is not None else None
d = hip.attn_desc(qd, kvd, kvd, o, lse, batch=B, hq=Hq, hkv=Hkv, sq=Sq, sk=Sk, hd=D, scale=D ** -0.5, causal=causal,
                  kv_start=kvsd, k_off=0, v_off=wkv)
hip.attention_fwd(d)
torch.cuda.synchronize()
print(case, "fwd", rel_err(o.float().cpu().view(B, Sq, Hq, D), ref.detach()), flush=True)
if mode == "dq":
    hip.attention_bwd(d, dod, dq)
else:
    hip.attention_bwd(d, dod, dq, dkv, dkv, dk_off=0, dv_off=wkv)
torch.cuda.synchronize()
print("dq", rel_err(dq.float().cpu().view(B, Sq, Hq, D), q.grad), flush=True)
if mode != "dq":
    print("dk", rel_err(dkv[:, :wkv].float().cpu().view(B, Sk, Hkv, D), k.grad),
          "dv", rel_err(dkv[:, wkv:].float().cpu().view(B, Sk, Hkv, D), v.grad), flush=True)
G = 1 << 20
for name, buf, view in (("o", obuf, o), ("lse", lbuf, lse), ("dq", dqbuf, dq), ("dkv", dkvbuf, dkv)):
    ok = bool((buf[:G] == 7).all() and (buf[G + view.numel():] == 7).all())
    print("guard", name, "intact" if ok else "OVERWRITTEN", flush=True)
