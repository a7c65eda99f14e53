"""Regression check of the GEMM across builds: one process per TREE (a checkout / `git archive` extract that holds its own
`desta2.5-audio_amd/` with a built `desta/lib/libdesta_hip.so`), plain NT GEMM with a bf16 store on the shapes that exposed round 4's
epilogue regression — an exact-4-round grid at K = 256 and K = 4096 (per-tile overhead vs main loop), the Whisper q|k|v / fc1 shapes
(K = 1280) and the LLM gate_up / down shapes.  Median of 5 x 10 launches, us per launch.

    git archive <old commit> desta2.5-audio_amd include | tar -x -C scratch/old && (cd scratch/old && python desta2.5-audio_amd/build.py)
    python tools/gemm_regression_check.py scratch/old ; python tools/gemm_regression_check.py .

An A/B of a run-time switch inside ONE binary cannot see what the switch's own code costs (profiles/r04_gemm_regression_bisect.log)."""
import os, sys
tree = sys.argv[1]
sys.path.insert(0, os.path.join(tree, "desta2.5-audio_amd"))
import torch
from desta import _hip as H

def t_us(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

out = []
for (M, N, K, tag) in ((4096, 16384, 256, "k256x4r"), (4096, 16384, 4096, "k4096x4r"), (12000, 3840, 1280, "wqkv"), (12000, 5120, 1280, "wfc1"), (5120, 28672, 4096, "gate_up"), (5120, 4096, 14336, "down")):
    A = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    B = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ts = sorted(t_us(lambda: H.gemm(A, B, C, M, N, K)) for _ in range(5))
    out.append(f"{tag} {ts[2]:.1f}us")
print(os.path.basename(tree.rstrip('/')).ljust(10), "  ".join(out), flush=True)
