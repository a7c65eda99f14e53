"""A/B of the de-synchronised first round of the 256x256 GEMM (desta_gemm_set_option(7, spread)): multi-round shapes of the
step with and without a residual epilogue, interleaved rounds in one process.  python tools/gemm_desync_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "desta2.5-audio_amd"))
import torch
from desta import _hip as H

SHAPES = [(5120, 28672, 4096, "llm gate_up", False), (4608, 14336, 4096, "llm d_act", False), (5120, 6144, 4096, "llm qkv", False),
          (4096, 128256, 4096, "lm_head", False), (12000, 5120, 1280, "whisper fc1 (bias+gelu)", True), (12000, 3840, 1280, "whisper qkv (bias)", True),
          (12000, 1280, 5120, "whisper fc2 (bias+res)", True), (12000, 1280, 1280, "whisper out (bias+res)", True)]
SPREADS = [0, 10, 20, 40]
for M, N, K, what, epi in SHAPES:
    A = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    B = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16 if "res" not in what else torch.float32, device="cuda")
    bias = torch.randn(N, device="cuda") if epi else None
    res = torch.randn(M, N, device="cuda") if "res" in what else None
    act = 1 if "gelu" in what else 0
    out = {s: [] for s in SPREADS}
    for r in range(6):
        for s in SPREADS:
            H.gemm_set_option(7, s)
            H.gemm_set_option(8, 300)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                H.gemm(A, B, C, M, N, K, bias=bias, residual=res, act=act)
            e1.record()
            torch.cuda.synchronize()
            if r:
                out[s].append(e0.elapsed_time(e1) * 1e3 / 5)
    H.gemm_set_option(7, 0)
    print(f"{what:26s} {M}x{N}x{K}: " + "  ".join(f"spread {s * 0.5:4.1f} us: {sorted(v)[len(v) // 2]:7.1f} us" for s, v in out.items()), flush=True)
