"""Reference point for the hand-written GEMM: the same NT shapes through torch.matmul (hipBLASLt / rocBLAS on ROCm), random bf16
operands, interleaved rounds in one process.  python tools/blas_compare.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "desta2.5-audio_amd"))
import torch
from desta import _hip as H

SHAPES = [(5120, 6144, 4096, "llm qkv"), (5120, 4096, 4096, "llm o_proj"), (5120, 28672, 4096, "llm gate_up"),
          (5120, 4096, 14336, "llm down"), (5120, 14336, 4096, "llm d_act"), (5120, 4096, 28672, "llm d_gu->h"),
          (4096, 128256, 4096, "lm_head (target rows)"), (12000, 3840, 1280, "whisper qkv"), (12000, 1280, 1280, "whisper out"),
          (12000, 5120, 1280, "whisper fc1"), (12000, 1280, 5120, "whisper fc2"), (48000, 2560, 1280, "qformer kv"),
          (2048, 5120, 1280, "qformer fc1"), (2048, 1280, 1280, "qformer out"), (8192, 8192, 8192, "8192^3")]


def t_us(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    print(f"{'shape':44s} {'this repo':>18s} {'torch.matmul':>18s}   (us, TFLOP/s; median of {rounds})")
    for M, N, K, what in SHAPES:
        A = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        B = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        Bt = B.t()
        mine, lib = [], []
        for r in range(rounds + 1):
            a = t_us(lambda: H.gemm(A, B, C, M, N, K), 3)
            b = t_us(lambda: torch.matmul(A, Bt, out=C), 3)
            if r:
                mine.append(a)
                lib.append(b)
        mine.sort()
        lib.sort()
        a, b = mine[len(mine) // 2], lib[len(lib) // 2]
        fl = 2.0 * M * N * K / 1e6
        print(f"{what:24s} {M:6d}x{N:6d}x{K:6d} {a:9.1f} {fl / a:7.0f}   {b:9.1f} {fl / b:7.0f}", flush=True)


if __name__ == "__main__":
    main()
