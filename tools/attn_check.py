"""Print fwd / dq / dk / dv relative errors of the attention kernels for every test case (no asserts)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("desta2.5-audio_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from desta import _hip as hip
from test_gpu_ops import ATTN_CASES, _attn_ref, bf, rel_err

for case in ATTN_CASES + [(8, 32, 8, 640, 640, 128, True, None), (2, 20, 20, 1500, 1500, 64, False, None)]:
    B, Hq, Hkv, Sq, Sk, D, causal, pad = case
    g = torch.Generator().manual_seed(Sq * 3 + Sk + D)
    wq, wkv = Hq * D, Hkv * D
    qb = bf(torch.randn(B * Sq, wq, generator=g))
    kvb = bf(torch.randn(B * Sk, 2 * wkv, generator=g))
    q = qb.float().reshape(B, Sq, Hq, D).clone().requires_grad_(True)
    k = kvb[:, :wkv].float().reshape(B, Sk, Hkv, D).clone().requires_grad_(True)
    v = kvb[:, wkv:].float().reshape(B, Sk, Hkv, D).clone().requires_grad_(True)
    kvs = torch.tensor(pad, dtype=torch.int32) if pad is not None else None
    big = B * Hq * Sq * Sk > 3e8
    if not big:
        ref = _attn_ref(q, k, v, D ** -0.5, causal, kvs)
        do = bf(torch.randn(B * Sq, wq, generator=g))
        ref.backward(do.float().view(B, Sq, Hq, D))
    else:
        do = bf(torch.randn(B * Sq, wq, generator=g))
    o = torch.zeros(B * Sq, wq, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, Hq, Sq, device="cuda")
    qd, kvd, dod = qb.cuda(), kvb.cuda(), do.cuda()          # keep the device buffers alive: the descriptor holds raw pointers
    d = hip.attn_desc(qd, kvd, kvd, o, lse, batch=B, hq=Hq, hkv=Hkv, sq=Sq, sk=Sk, hd=D,
                      scale=D ** -0.5, causal=causal, kv_start=kvs.cuda() if kvs is not None else None, k_off=0, v_off=wkv)
    hip.attention_fwd(d)
    torch.cuda.synchronize()
    dq = torch.zeros(B * Sq, wq, dtype=torch.bfloat16, device="cuda")
    dkv = torch.zeros(B * Sk, 2 * wkv, dtype=torch.bfloat16, device="cuda")
    hip.attention_bwd(d, dod, dq, dkv, dkv, dk_off=0, dv_off=wkv)
    torch.cuda.synchronize()
    if big:
        # determinism + finiteness only (reference too large for the CPU)
        dq2 = torch.zeros_like(dq)
        dkv2 = torch.zeros_like(dkv)
        hip.attention_bwd(d, dod, dq2, dkv2, dkv2, dk_off=0, dv_off=wkv)
        print(case, "finite", bool(torch.isfinite(dq.float()).all() and torch.isfinite(dkv.float()).all()),
              "deterministic", torch.equal(dq, dq2) and torch.equal(dkv, dkv2), flush=True)
        continue
    print(case, "fwd %.4f dq %.4f dk %.4f dv %.4f" % (
        rel_err(o.float().cpu().view(B, Sq, Hq, D), ref.detach()),
        rel_err(dq.float().cpu().view(B, Sq, Hq, D), q.grad),
        rel_err(dkv[:, :wkv].float().cpu().view(B, Sk, Hkv, D), k.grad),
        rel_err(dkv[:, wkv:].float().cpu().view(B, Sk, Hkv, D), v.grad)), flush=True)
