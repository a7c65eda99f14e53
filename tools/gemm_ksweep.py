"""Per-tile cost model of the 256x256 GEMM kernel: time per round (256 tiles = one per CU) as a function of K
on a grid of exactly R full rounds -> slope = time per 64-wide K-step, intercept = per-tile overhead
(prologue + epilogue + block turnover).  python tools/gemm_ksweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "desta2.5-audio_amd"))
import torch
from desta import _hip as H


def t_us(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = "cuda"
    for (M, N, rounds) in ((4096, 4096, 1), (4096, 16384, 4), (4096, 32768, 8)):
        for epi in ("plain", "bias+gelu", "residual"):
            pts = []
            for K in (256, 512, 1024, 2048, 4096, 8192):
                A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
                B = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
                C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                bias = torch.randn(N, device=dev) if epi == "bias+gelu" else None
                res = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == "residual" else None
                H.gemm_force_variant(6)
                us = t_us(lambda: H.gemm(A, B, C, M, N, K, bias=bias, residual=res, act=1 if bias is not None else 0))
                H.gemm_force_variant(0)
                pts.append((K // 64, us / rounds))
            # least squares fit  t = a + b * ksteps  on the 4 largest K
            xs = [p[0] for p in pts[2:]]
            ys = [p[1] for p in pts[2:]]
            n = len(xs)
            mx, my = sum(xs) / n, sum(ys) / n
            b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
            a = my - b * mx
            print(f"M={M} N={N} rounds={rounds} {epi:10s} per-round us by K: " + " ".join(f"{k*64}:{t:.1f}" for k, t in pts) +
                  f"  | fit: {b:.3f} us/K-step ({256 * 2 * 256 * 256 * 64 / (b * 1e-6) / 1e12:.0f} TF/s main loop), overhead {a:.1f} us/tile", flush=True)


if __name__ == "__main__":
    main()
