"""Weight-gradient GEMM of the ORCA cross-attention at full size, dW [N, Kin] = dY[M, N]^T X[M, Kin]: (a) both operands in transposed
storage on the 128x128 kernel (no copies) against (b) two explicit bf16 transposes + the 256x256 NT kernel.
`python tools/orca_dw_bench.py` -> one line per shape (us per dW, TFLOP/s)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "desta2.5-audio_amd"))
from desta import _hip as H                                                                  # noqa: E402

BF16 = torch.bfloat16


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = "cuda:0"
    for name, N, Kin, M in (("llama q/out", 4096, 4096, 5120), ("llama k|v", 8192, 4096, 3072), ("llama gate0", 1024, 4096, 5120),
                            ("qwen3-4b q (padded)", 4096, 2560, 5120), ("qwen3-0.6b q", 1024, 1024, 5120)):
        dY = torch.randn(M, N, device=dev).to(BF16)
        X = torch.randn(M, Kin, device=dev).to(BF16)
        gw, gw2 = torch.empty(N, Kin, device=dev), torch.empty(N, Kin, device=dev)
        tA, tB = torch.empty(N, M, dtype=BF16, device=dev), torch.empty(Kin, M, dtype=BF16, device=dev)

        def direct():
            H.gemm(dY, X, gw, N, Kin, M, trans_a=True, trans_b=True, lda=N, ldb=Kin)

        def via_t():
            H.transpose_to_bf16(dY, M, N, tA, M)
            H.transpose_to_bf16(X, M, Kin, tB, M)
            H.gemm(tA, tB, gw2, N, Kin, M)

        def via_t_shared_x():                                                # X^T already there (shared by two dW of a layer)
            H.transpose_to_bf16(dY, M, N, tA, M)
            H.gemm(tA, tB, gw2, N, Kin, M)
        a, b, c = t(direct), t(via_t), t(via_t_shared_x)
        err = float((gw - gw2).abs().max() / gw.abs().max())
        fl = 2.0 * N * Kin * M / 1e6
        print(f"{name:22s} N {N:5d} Kin {Kin:5d} M {M:5d}: transposed storage {a:7.1f} us ({fl / a:6.0f} TF/s) | 2 transposes + NT256 {b:7.1f} us ({fl / b:6.0f}) | "
              f"1 transpose + NT256 {c:7.1f} us ({fl / c:6.0f}) | max diff {err:.1e}")


if __name__ == "__main__":
    main()
