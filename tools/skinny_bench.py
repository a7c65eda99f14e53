"""Weight-streaming rate of the skinny (M <= 16) GEMM per shape and kernel variant.  Each timed launch reads a
DIFFERENT weight buffer (rotating set > 1 GiB) so neither L2 nor the 256 MB MALL can serve the weights.
  python tools/skinny_bench.py [--m 8]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "desta2.5-audio_amd"))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=8)
    ap.add_argument("--variants", default="0,-256,-768,-1024,164,322")
    a = ap.parse_args()
    from desta import _hip as H
    M = a.m
    shapes = [(4096, 4096), (6144, 4096), (28672, 4096), (4096, 14336), (128256, 4096)]
    for N, K in shapes:
        nbuf = max(2, int(1.5 * 2**30 / (N * K * 2)) + 1)
        Ws = [(torch.randn(N, K, device="cuda", dtype=torch.float32) * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ref = (x.float() @ Ws[0].float().T)
        line = [f"N={N:6d} K={K:5d}"]
        for v in [int(t) for t in a.variants.split(",")]:
            H.gemm_set_option(3, -v if v < 0 else 512)            # negative code: 16x2 with a persistent grid of -v blocks
            H.gemm_set_option(2, 0 if v < 0 else v)
            H.gemm(x, Ws[0], out, M, N, K)
            err = float((out.float() - ref).abs().max() / ref.abs().max())
            assert err < 2e-2, (v, err)
            for i in range(nbuf):
                H.gemm(x, Ws[i], out, M, N, K)
            reps = max(2 * nbuf, 20)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps):
                H.gemm(x, Ws[i % nbuf], out, M, N, K)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            line.append(f"v{v}: {us:7.1f}us {N * K * 2 / us / 1e6:5.2f}TB/s")
        H.gemm_set_option(2, 0)
        H.gemm_set_option(3, 512)
        print("  ".join(line), flush=True)
        del Ws


if __name__ == "__main__":
    main()
