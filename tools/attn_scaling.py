"""Forward attention kernels over sequence length: separates the loop rate from the per-block overhead (8-wave lockstep /
8-wave with the half-tile stagger / 4-wave).  python tools/attn_scaling.py"""
import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'desta2.5-audio_amd'))
from desta import _hip as hip
def t_us(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for (B, Hq, Hkv, S, D, causal) in [(8,32,8,640,128,True),(8,32,8,1280,128,True),(8,32,8,2560,128,True),(8,32,8,640,128,False),(8,32,8,1280,128,False),
                                   (8,32,32,640,128,False),(8,20,20,1500,64,False),(8,20,20,3000,64,False),(8,32,8,640,64,True)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    wq, wkv = Hq * D, Hkv * D
    qkv = torch.randn(B * S, wq + 2 * wkv, generator=g, device="cuda").to(torch.bfloat16)
    o = torch.zeros(B * S, wq, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, Hq, S, device="cuda")
    d = hip.attn_desc(qkv, qkv, qkv, o, lse, batch=B, hq=Hq, hkv=Hkv, sq=S, sk=S, hd=D, scale=D ** -0.5, causal=causal, q_off=0, k_off=wq, v_off=wq + wkv)
    res = []
    for opts in ((1,0,0),(1,0,1),(0,0,0)):      # (option 0: 8-wave, unused, option 2: stagger)
        hip.attention_set_option(0, opts[0]); hip.attention_set_option(2, opts[2])
        res.append(t_us(lambda: hip.attention_fwd(d)))
    hip.attention_set_option(0, 1); hip.attention_set_option(2, 0)
    fl = 4.0 * B * Hq * S * S * D * (0.5 if causal else 1.0)
    print(f"B{B} Hq{Hq} Hkv{Hkv} S{S} D{D} causal={causal}: 8w-nostagger {res[0]:8.1f} us ({fl/res[0]/1e6:5.0f} TF)  8w-stagger {res[1]:8.1f}  4w {res[2]:8.1f} ({fl/res[2]/1e6:5.0f} TF)", flush=True)
