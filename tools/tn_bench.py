"""A/B: transposed-storage GEMM operands vs materialised transposes + NT GEMM on the Q-Former backward shapes."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "desta2.5-audio_amd"))
import torch
from desta import _hip as H


def t_us(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


R = 2048
for (N, Kin) in ((3840, 1280), (1280, 1280), (3072, 1280), (1280, 3072), (4096, 1280)):
    M = R if N != 4096 else 512
    dY = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    X = torch.randn(M, Kin, device="cuda").to(torch.bfloat16)
    W = torch.randn(N, Kin, device="cuda").to(torch.bfloat16)
    tA = torch.empty(N, M, dtype=torch.bfloat16, device="cuda")
    tB = torch.empty(Kin, M, dtype=torch.bfloat16, device="cuda")
    WT = torch.empty(Kin, N, dtype=torch.bfloat16, device="cuda")
    gw = torch.empty(N, Kin, dtype=torch.float32, device="cuda")
    dx = torch.empty(M, Kin, dtype=torch.float32, device="cuda")

    def dw_nt():
        H.transpose_to_bf16(dY, M, N, tA, M)
        H.transpose_to_bf16(X, M, Kin, tB, M)
        H.gemm(tA, tB, gw, N, Kin, M)

    def dw_tn():
        H.gemm(dY, X, gw, N, Kin, M, trans_a=True, trans_b=True)
    H.transpose_to_bf16(W, N, Kin, WT, N)

    def dx_nt():
        H.gemm(dY, WT, dx, M, Kin, N)

    def dx_tb():
        H.gemm(dY, W, dx, M, Kin, N, trans_b=True)
    only = t_us(lambda: H.gemm(tA, tB, gw, N, Kin, M))
    print(f"N={N} Kin={Kin} M={M}: dW transposes+NT {t_us(dw_nt):6.1f} us (NT gemm alone {only:6.1f})  TN {t_us(dw_tn):6.1f} us | "
          f"dX NT {t_us(dx_nt):6.1f} us  trans_b {t_us(dx_tb):6.1f} us", flush=True)

# transposes of the Q-Former backward (bf16): the tall d(K|V) one and a small dY
for rows, cols in ((48000, 2560), (2048, 3840), (2048, 1280)):
    x = torch.randn(rows, cols, device="cuda").to(torch.bfloat16)
    o = torch.empty(cols, rows, dtype=torch.bfloat16, device="cuda")
    us = t_us(lambda: H.transpose_to_bf16(x, rows, cols, o, rows))
    print(f"transpose [{rows}, {cols}] bf16: {us:6.1f} us = {2 * rows * cols * 2 / us / 1e6:5.2f} TB/s (read + write)", flush=True)
