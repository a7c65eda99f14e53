"""Per-(shape, epilogue) table of every GEMM call of ONE un-overlapped training step of the headline config: calls per step, us per
call (HIP events around the call, incl. a split-K fix-up launch), TFLOP/s, ms per step — sorted by time.  Finds the shapes /
epilogue modes that run below the kernel's usual rate in situ.   python tools/gemm_shapes_in_step.py [config]"""
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "desta2.5-audio_amd"))
import torch  # noqa: E402
from desta import _hip as H  # noqa: E402


def main():
    from desta.models.modeling_desta25 import DeSTA25AudioModel, DeSTA25Config
    from desta.synthetic import FULL_CONFIGS, RandomWeights, synthetic_inputs, synthetic_waveform
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments
    name = sys.argv[1] if len(sys.argv) > 1 else "desta25_llama31-8B_Qformer6L"
    dev = torch.device("cuda:0")
    cfg = DeSTA25Config(**FULL_CONFIGS[name])
    model = DeSTA25AudioModel(cfg, weights=RandomWeights(cfg, dev, seed=0), device=dev)
    tr = DeSTA25Trainer(model, args=TrainingArguments(learning_rate=1e-4, warmup_steps=0, max_steps=100, logging_steps=10 ** 9, overlap_comm=False,
                                                      overlap_encoder=False))
    B = 8
    batch = synthetic_inputs(cfg, B, 64, 512, dev, seed=3)
    batch["batch_features"] = H.logmel(synthetic_waveform(B, dev, seed=3), cfg.encoder_config.num_mel_bins)
    model.connector.kv_side, model.connector.overlap_dw = False, False
    for _ in range(3):
        tr.training_step(batch)
    torch.cuda.synchronize()
    rec = []
    orig = H.gemm

    def wrapped(A, Bm, C, M, N, K, **kw):
        mode = "+".join(k for k in ("bias", "residual", "preact", "rope", "aux") if kw.get(k) is not None)
        if kw.get("act"):
            mode += f"+act{kw['act']}"
        if C.dtype == torch.float32:
            mode += "+f32out"
        if kw.get("trans_a") or kw.get("trans_b"):
            mode += "+T"
        if kw.get("batch", 1) > 1:
            mode += f"+batch{kw['batch']}"
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        orig(A, Bm, C, M, N, K, **kw)
        b.record()
        rec.append(((M, N, K, mode or "plain", H.lib.desta_gemm_last_kernel()), 2.0 * M * N * K * kw.get("batch", 1), a, b))
    H.gemm = wrapped
    import desta.models.modeling_desta25 as MM
    MM.H.gemm = wrapped
    steps = 3
    for _ in range(steps):
        tr.training_step(batch)
    torch.cuda.synchronize()
    H.gemm = orig
    agg = OrderedDict()
    for key, fl, a, b in rec:
        n, f, ms = agg.get(key, (0, 0.0, 0.0))
        agg[key] = (n + 1, f + fl, ms + a.elapsed_time(b))
    tot = sum(v[2] for v in agg.values()) / steps
    print(f"{name}: {len(rec) // steps} GEMM calls per step, {tot:.2f} ms per step in GEMM calls (un-overlapped, events per call)")
    print(f"{'M':>6} {'N':>7} {'K':>7}  kernel  {'epilogue':28s} calls/step   us/call   TFLOP/s   ms/step")
    for (M, N, K, mode, kern), (n, f, ms) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        print(f"{M:6d} {N:7d} {K:7d}  {('128', '256', 'skinny')[kern - 1] if kern in (1, 2, 3) else kern:>6}  {mode:28s} {n / steps:8.1f} {1e3 * ms / n:9.1f} {f / (ms * 1e-3) / 1e12:9.0f} {ms / steps:9.3f}")


if __name__ == "__main__":
    main()
