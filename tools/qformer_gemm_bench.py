"""Time the GEMM shapes of the Q-Former connector / projector (the launches that fall to the 128x128 kernel)
one by one: normal and transposed-storage (dW) forms, back-to-back launches between two events.
  python tools/qformer_gemm_bench.py [variant] [ring]      (ring: option 6 of desta_gemm_set_option, 0 / 1 / 2)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "desta2.5-audio_amd"))
import torch
from desta import _hip as H

# (M, N, K, trans_a, trans_b, what): out[M,N] = A[M,K] B[N,K]^T; trans_* = the operand is stored [K, M] / [K, N]
SHAPES = [
    (2048, 3840, 1280, 0, 0, "self qkv fwd"), (2048, 1280, 1280, 0, 0, "out / cross-q fwd"), (2048, 5120, 1280, 0, 0, "fc1 fwd"),
    (2048, 1280, 5120, 0, 0, "fc2 fwd / d_fc1->x"), (2048, 1280, 3840, 0, 0, "d_qkv->x"),
    (1280, 1280, 2048, 1, 1, "dW out (X^T dY)"), (3840, 1280, 2048, 1, 1, "dW qkv"), (5120, 1280, 2048, 1, 1, "dW fc1"),
    (1280, 5120, 2048, 1, 1, "dW fc2"), (512, 4096, 1280, 0, 0, "projector fwd"), (512, 1280, 4096, 0, 0, "projector dX"),
    (4096, 1280, 512, 1, 1, "projector dW"),
]


def main():
    variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    dev = "cuda"
    H.gemm_force_variant(variant)
    if len(sys.argv) > 2:
        H.gemm_set_option(6, int(sys.argv[2]))
    tot = 0.0
    for M, N, K, ta, tb, what in SHAPES:
        A = (torch.rand((K, M) if ta else (M, K), device=dev) * 2 - 1).to(torch.bfloat16)
        B = (torch.rand((K, N) if tb else (N, K), device=dev) * 2 - 1).to(torch.bfloat16)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        best = 1e9
        for r in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                H.gemm(A, B, C, M, N, K, trans_a=bool(ta), trans_b=bool(tb))
            e1.record()
            torch.cuda.synchronize()
            if r:
                best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
        tot += best
        print(f"{what:22s} {M:5d}x{N:5d}x{K:5d} ta={ta} tb={tb}  {best:7.1f} us  {2.0 * M * N * K / best / 1e6:6.0f} TF/s  kernel {H.lib.desta_gemm_last_kernel()}", flush=True)
    print(f"sum {tot:.1f} us")
    H.gemm_force_variant(0)


if __name__ == "__main__":
    main()
