"""A/B of `desta_gemm_set_option(10, .)` — the two-phase 256x256 GEMM stopping its half-tile stream at the last K-tile (1) against
re-loading dead LDS slots (0) — on the shapes of the step, interleaved rounds in one process, random bf16 operands.
    python tools/gemm_tailskip_bench.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "desta2.5-audio_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from desta import _hip as H
from gemm_bench import SHAPES


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    print(f"{'shape':34s} dummy-loads TF/s   tail-skip TF/s   ratio")
    for M, N, K, what in SHAPES:
        A = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        B = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        res = {0: [], 1: []}
        for r in range(rounds + 1):
            for opt in (0, 1):
                H.gemm_set_option(10, opt)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    H.gemm(A, B, C, M, N, K)
                e1.record()
                e1.synchronize()
                if r:
                    res[opt].append(2.0 * M * N * K * 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        med = {o: sorted(v)[len(v) // 2] for o, v in res.items()}
        print(f"{what + f' {M}x{N}x{K}':34s} {med[0]:12.0f} {med[1]:16.0f}   {med[1] / med[0]:.3f}")
    H.gemm_set_option(10, 1)


if __name__ == "__main__":
    main()
