"""A/B the GEMM tile variants on the shapes of the DeSTA2.5 step (interleaved rounds, one process,
random bf16 operands).  python tools/gemm_bench.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "desta2.5-audio_amd"))
import torch
from desta import _hip as H

SHAPES = [  # (M, N, K, what)
    (5120, 6144, 4096, "llm qkv"), (5120, 4096, 4096, "llm o_proj"), (5120, 28672, 4096, "llm gate_up"),
    (5120, 4096, 14336, "llm down"), (5120, 14336, 4096, "llm d_act"), (5120, 4096, 28672, "llm d_gu->h"),
    (5120, 4096, 6144, "llm d_qkv->h"), (5120, 128256, 4096, "lm_head"), (5120, 4096, 128256, "d_logits->h"),
    (12000, 3840, 1280, "whisper qkv"), (12000, 1280, 1280, "whisper out"), (12000, 5120, 1280, "whisper fc1"),
    (12000, 1280, 5120, "whisper fc2"), (48000, 2560, 1280, "qformer kv proj"), (2560, 1280, 48000, "qformer dW kv"),
    (2048, 3840, 1280, "qformer qkv"), (2048, 1280, 3072, "qformer ffn out"), (4096, 4096, 4096, "4096^3"),
    (8192, 8192, 8192, "8192^3"),
]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2]
    dev = "cuda"
    print(f"{'shape':34s} " + " ".join(f"v{v:>2d} TF/s (med,max)" for v in variants))
    flt = sys.argv[3] if len(sys.argv) > 3 else ""
    for M, N, K, what in SHAPES:
        if flt and not any(f in what for f in flt.split(",")):
            continue
        A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
        B = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        res = {v: [] for v in variants}
        for r in range(rounds + 1):
            for v in variants:
                H.gemm_force_variant(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 3
                e0.record()
                for _ in range(reps):
                    H.gemm(A, B, C, M, N, K)
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    res[v].append(2.0 * M * N * K * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        H.gemm_force_variant(0)
        cols = []
        for v in variants:
            xs = sorted(res[v])
            cols.append(f"{xs[len(xs) // 2]:8.0f} {xs[-1]:8.0f}  ")
        print(f"{what:18s} {M:6d}x{N:6d}x{K:6d} " + " ".join(cols), flush=True)


if __name__ == "__main__":
    main()
