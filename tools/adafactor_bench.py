"""Fused clip + Adafactor on the real connector arena (131.54 M fp32): HIP-event time per call and the byte rates the
roofline section quotes (algorithmic 12 N; this implementation's HBM traffic 16 N).
  python tools/adafactor_bench.py [--iters 50]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "desta2.5-audio_amd"))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    from desta.models.modeling_desta25 import DeSTA25Config, connector_param_shapes
    from desta.optim import FusedAdafactor, ParamArena
    from desta.synthetic import FULL_CONFIGS
    cfg = DeSTA25Config(**FULL_CONFIGS["desta25_llama31-8B_Qformer6L"])
    arena = ParamArena(list(connector_param_shapes(cfg).items()), "cuda")
    arena.params.normal_(0, 0.02)
    opt = FusedAdafactor(arena)
    N = arena.numel
    flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda")      # 1 GiB: evicts the Infinity Cache between calls
    times = {"cold": [], "warm": []}
    for mode in ("cold", "warm"):
        for i in range(a.iters):
            arena.grads.normal_(0, 1e-3 * (1 + i % 3))
            if mode == "cold":
                flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            opt.step(1e-4)
            e1.record()
            torch.cuda.synchronize()
            times[mode].append(e0.elapsed_time(e1))
    for mode, ts in times.items():
        ts = sorted(ts[5:])
        med = ts[len(ts) // 2]
        print(f"{mode}: median {1e3 * med:.0f} us (min {1e3 * ts[0]:.0f}, max {1e3 * ts[-1]:.0f}) over {len(ts)} calls | N = {N / 1e6:.2f} M floats | "
              f"algorithmic 12 N = {12 * N / 1e9:.3f} GB -> {12 * N / med / 1e6:.0f} GB/s ({12 * N / med / 1e6 / 8000:.3f} of 8 TB/s) | "
              f"moved 16 N = {16 * N / 1e9:.3f} GB -> {16 * N / med / 1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
