#!/usr/bin/env python
"""bench.py — train steps/s of the MI355X-native DeSTA2.5-Audio hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one per-GPU batch (B=8 synthetic 30 s clips + 640
tokens): log-mel -> Whisper-large-v3 encoder -> Q-Former 6L -> projector -> splice -> Llama-3.1-8B ->
CE -> backward (dX through the LLM, dW+dX through the connector) -> mean all-reduce of the flat
gradient arena over RCCL (N>1) -> global-norm clip -> Adafactor -> LR schedule.  Inputs (waveforms,
token ids) are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "desta2.5-audio_amd"))

import torch
import torch.distributed as dist

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md §Chip-level parameters
HBM_PEAK_GBS = 8000.0               # HBM3E, same guide


def cpu_baseline_debug(threads, steps=20):
    """the debug config (tiny Whisper-like encoder + tiny causal LM, Q-Former 2L) timed FULLY —
    every step is the whole HF-Trainer-ordered step of the oracle (forward under the autocast(bf16) policy -> loss -> backward ->
    clip 1.0 -> Adafactor -> schedule) on 30 s clips (3000 mel frames), B = 2, 1 warm-up + `steps` timed steps."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import desta_oracle as O
    torch.set_num_threads(threads)
    d = O.tiny_dims(False)
    d.enc_T = 1500
    w = O.init_weights(d, seed=0)
    st = O.adafactor_init([w[n] for n in O.trainable_names(d)])
    pool = [O.synthetic_batch(d, B=2, S_ctx=8, S_tgt=24, seed=10 + i) for i in range(4)]
    ts = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        O.train_step(w, d, pool[i % 4], st, O.linear_warmup_lr(i, 1e-4, 5, steps + 1), autocast=True)
        ts.append(time.perf_counter() - t0)
    ts = ts[1:]
    mean = sum(ts) / len(ts)
    sd = (sum((t - mean) ** 2 for t in ts) / max(1, len(ts) - 1)) ** 0.5
    return {"value": 1.0 / mean, "unit": "steps/s", "steps": steps, "ms_per_step": 1e3 * mean, "ms_per_step_sd": 1e3 * sd,
            "precision": "autocast(bf16) policy", "workload": "desta25_debug: B=2 x 30 s clips, S=96, 4-layer d=128 encoder, Q-Former 2L, 2-layer h=256 LM"}


def cpu_baseline(cfg, B, S_ctx, S_tgt, threads):
    """CPU 'port' baseline: the oracle (oracle/desta_oracle.py, plain fp32 PyTorch restatement of the
    reference step) timed on this box's host cores on a BOUNDED sample of the same workload — one layer
    of each stack at true width at B=1 — and scaled by layer counts and batch to one full step."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import desta_oracle as O
    torch.set_num_threads(threads)
    c, e = cfg.llm_config, cfg.encoder_config
    sc = c.rope_scaling
    d = O.Dims(n_mels=e.num_mel_bins, enc_d=e.d_model, enc_layers=1, enc_heads=e.encoder_attention_heads, enc_ffn=e.encoder_ffn_dim,
               enc_T=e.max_source_positions, taps=(0,), qf_layers=1, qf_inter=cfg.qformer_intermediate_size, prompt_size=cfg.prompt_size,
               llm_h=c.hidden_size, llm_layers=1, llm_hq=c.num_attention_heads, llm_hkv=c.num_key_value_heads, llm_hd=c.head_dim,
               llm_inter=c.intermediate_size, vocab=c.vocab_size, rms_eps=c.rms_norm_eps, rope_theta=c.rope_theta,
               rope_llama3=(sc["factor"], sc["low_freq_factor"], sc["high_freq_factor"], sc["original_max_position_embeddings"]) if sc else None,
               qk_norm=c.qk_norm)
    w = O.init_weights(d, seed=0)
    batch = O.synthetic_batch(d, B=1, S_ctx=S_ctx, S_tgt=S_tgt, seed=1)
    names = O.trainable_names(d)

    def timed(fn, reps=1):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps
    wave = (0.1 * torch.randn(1, 480000)).clamp(-1, 1)
    t_mel = timed(lambda: O.logmel(wave, d.n_mels))
    with torch.no_grad():
        mel = batch["batch_features"]
        t_stem = timed(lambda: O.whisper_stem(w, d, mel))
        x = O.whisper_stem(w, d, mel)
        t_enc = timed(lambda: O.whisper_layer(w, d, 0, x))
    for n in names:
        w[n].requires_grad_(True)

    def qf():
        y = O.qformer_layer(w, d, 0, w[O.CON + "layer_prompts.0"].expand(1, -1, -1), x)
        y.sum().backward()
    t_qf = timed(qf)
    af = torch.randn(1, d.prompt_size, d.llm_h, requires_grad=True)

    def llm(layers):
        dd = O.Dims(**{**d.__dict__, "llm_layers": layers})
        xe = O.embed_splice(w, dd, batch["input_ids"], af, batch["batch_transcription_ids"], batch["batch_start_positions"])
        loss = O.causal_lm_loss(O.llm_forward(w, dd, xe, batch["attention_mask"]), batch["labels"])
        loss.backward()
    t_l1 = timed(lambda: llm(1))
    t_l0 = timed(lambda: llm(0))
    t_layer, t_head = max(t_l1 - t_l0, 1e-6), t_l0
    # the same LLM layer under the reference's autocast(bf16) policy (BASELINE.md §3 names it): reported beside fp32; the
    # FASTER of the two is the baseline (CPU bf16 GEMMs are slower than fp32 without AMX / AVX512-BF16)
    with O.autocast_bf16():
        t_l1_ac = timed(lambda: llm(1))
        t_l0_ac = timed(lambda: llm(0))
    t_layer_ac = max(t_l1_ac - t_l0_ac, 1e-6)
    nt = len(cfg.target_layer_ids)
    fixed = t_mel + t_stem + e.encoder_layers * t_enc + nt * cfg.qformer_num_hidden_layers * t_qf
    full_fp32 = B * (fixed + c.num_hidden_layers * t_layer + t_head)
    full_mix = B * (fixed + c.num_hidden_layers * min(t_layer, t_layer_ac) + min(t_head, t_l0_ac))
    sample = (f"oracle fp32 on {threads} threads, B=1: log-mel {t_mel:.2f}s, conv stem {t_stem:.2f}s, 1 Whisper layer fwd {t_enc:.2f}s, "
              f"1 Q-Former layer fwd+bwd on one tap {t_qf:.2f}s, 1 LLM layer fwd+bwd {t_layer:.2f}s (autocast-bf16 policy: {t_layer_ac:.2f}s), "
              f"final norm+lm_head+CE fwd+bwd {t_head:.2f}s (autocast {t_l0_ac:.2f}s); `value` is the fp32 extrapolation "
              f"scaled to B={B}, {e.encoder_layers}+{nt}x{cfg.qformer_num_hidden_layers}+{c.num_hidden_layers} layers (optimizer/clip time excluded: <1% of the step)")
    sample += ("; the fp32 one-layer extrapolation is the one validated by a real full-depth fp32 B=1 step (python bench.py --cpu-full-step, "
               "log committed as profiles/r02_cpu_full_step_B1.log: 34.3 s per sample); `value_fastest_precision_mix` takes the faster of "
               "fp32 / autocast(bf16)-policy per component and is NOT validated by a full step")
    return {"value": 1.0 / full_fp32, "unit": "steps/s", "cores": threads, "kind": "port", "precision": "fp32", "sample": sample,
            "value_fastest_precision_mix": 1.0 / full_mix}


def cpu_full_step(cfg, S_ctx, S_tgt, threads, steps=2):
    """One REAL full-depth oracle step at true shapes, B = 1 (32 Whisper + 4 x 6 Q-Former + 32 LLM layers, lm_head, CE, backward,
    clip, Adafactor), fp32: validates the one-layer extrapolation of `cpu_baseline`.  Minutes of CPU time and ~45 GB of host
    memory: not part of the default run."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import desta_oracle as O
    torch.set_num_threads(threads)
    c, e = cfg.llm_config, cfg.encoder_config
    sc = c.rope_scaling
    d = O.Dims(n_mels=e.num_mel_bins, enc_d=e.d_model, enc_layers=e.encoder_layers, enc_heads=e.encoder_attention_heads, enc_ffn=e.encoder_ffn_dim,
               enc_T=e.max_source_positions, taps=tuple(cfg.target_layer_ids), qf_layers=cfg.qformer_num_hidden_layers,
               qf_inter=cfg.qformer_intermediate_size, prompt_size=cfg.prompt_size,
               llm_h=c.hidden_size, llm_layers=c.num_hidden_layers, llm_hq=c.num_attention_heads, llm_hkv=c.num_key_value_heads, llm_hd=c.head_dim,
               llm_inter=c.intermediate_size, vocab=c.vocab_size, rms_eps=c.rms_norm_eps, rope_theta=c.rope_theta,
               rope_llama3=(sc["factor"], sc["low_freq_factor"], sc["high_freq_factor"], sc["original_max_position_embeddings"]) if sc else None,
               qk_norm=c.qk_norm, tie_embeddings=c.tie_word_embeddings)
    t0 = time.perf_counter()
    w = O.init_weights(d, seed=0)
    print(f"[cpu-full-step] weights ({sum(v.numel() for v in w.values()) / 1e9:.2f} B fp32 values) in {time.perf_counter() - t0:.0f}s", flush=True)
    st = O.adafactor_init([w[n] for n in O.trainable_names(d)])
    batch = O.synthetic_batch(d, B=1, S_ctx=S_ctx, S_tgt=S_tgt, seed=1)
    wave = (0.1 * torch.randn(1, 480000)).clamp(-1, 1)
    ts = []
    for i in range(steps):
        t0 = time.perf_counter()
        batch["batch_features"] = O.logmel(wave, d.n_mels)
        loss, _, _, _ = O.train_step(w, d, batch, st, 1e-4)
        ts.append(time.perf_counter() - t0)
        print(f"[cpu-full-step] step {i}: {ts[-1]:.1f}s, loss {float(loss):.4f}", flush=True)
    return {"seconds_per_sample_step": ts[-1], "first_step_seconds": ts[0], "steps_per_s_at_B8_equivalent": 1.0 / (8 * ts[-1]),
            "cores": threads, "precision": "fp32"}


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(a, argv):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks as a FRESH child process —
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>`, the driver's own command line —
    BEFORE this process has touched the GPU, relay its output (rank 0 prints the JSON line) and exit with its code.  A request
    for more ranks than the node has GPUs exits non-zero instead of quietly running fewer."""
    import subprocess
    if not a.single_device and not a.launch_check:
        have = torch.cuda.device_count()                 # counting devices does not initialise the GPU
        if have < a.gpus:
            print(f"[bench] --gpus {a.gpus} but this node exposes {have} GPU(s); refusing to report a smaller run "
                  "(use --backend gloo --single-device for a one-GPU rehearsal)", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    print(f"[bench] launching {a.gpus} ranks: {' '.join(cmd[1:9])} ...", file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def launch_check(a, world, rank):
    """Plumbing-only run of the N-rank path (no device work): rendezvous, one SUM all-reduce, barrier, MAX over ranks, rank 0
    prints a line with `n_gpus`.  What the CPU test of the self-launch drives (`--launch-check --backend gloo`)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(a.backend if a.backend != "nccl" else "gloo")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "launch-check (no device work)", "value": None, "n_gpus": dist.get_world_size(),
                          "ranks_sum": float(t), "requested_gpus": a.gpus, "backend": dist.get_backend()}))
    dist.destroy_process_group()


def box_calibration(H, dev, warm_seconds=2.0, launches=20):
    """The same fixed work on every box, measured in this process right before the timed region, so that a reader can
    separate the BOX (clock under the power cap: the same binary spreads +-2.5 % over boxes) from the CODE when
    comparing bench lines across boxes and rounds: this library's bf16 GEMM on 8192^3 and on the gate_up shape
    (5120 x 28672 x 4096), random operands, HIP events around `launches` back-to-back calls after a warm loop, and one
    streaming device-to-device copy of 1 GiB."""
    out = {}
    g = torch.Generator(device=dev).manual_seed(99)

    def rnd(r, c):
        return (torch.rand(r, c, generator=g, device=dev, dtype=torch.float32) * 2 - 1).mul_(c ** -0.5).to(torch.bfloat16)

    def timed(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / n
    A, Bm, C = rnd(8192, 8192), rnd(8192, 8192), torch.empty(8192, 8192, dtype=torch.bfloat16, device=dev)
    t_end = time.perf_counter() + warm_seconds
    while time.perf_counter() < t_end:                   # clocks settle under sustained MFMA load
        for _ in range(10):
            H.gemm(A, Bm, C, 8192, 8192, 8192)
        torch.cuda.synchronize()
    dt = timed(lambda: H.gemm(A, Bm, C, 8192, 8192, 8192), launches)
    out["gemm_8192cubed_tflops"] = 2 * 8192 ** 3 / dt / 1e12
    del A, Bm, C
    X, W, Y = rnd(5120, 4096), rnd(28672, 4096), torch.empty(5120, 28672, dtype=torch.bfloat16, device=dev)
    for _ in range(5):
        H.gemm(X, W, Y, 5120, 28672, 4096)
    dt = timed(lambda: H.gemm(X, W, Y, 5120, 28672, 4096), launches)
    out["gemm_gate_up_tflops"] = 2 * 5120 * 28672 * 4096 / dt / 1e12
    del X, W, Y
    src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_(generator=g)
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    dt = timed(lambda: dst.copy_(src), 10)
    out["copy_1GiB_GBps"] = 2 * src.numel() * 4 / dt / 1e9                   # bytes read + bytes written
    del src, dst
    pr = torch.cuda.get_device_properties(dev)
    out["device"] = {"name": pr.name, "arch": getattr(pr, "gcnArchName", None), "uuid": str(getattr(pr, "uuid", "")) or None,
                     "pci_bus_id": getattr(pr, "pci_bus_id", None), "compute_units": pr.multi_processor_count}
    out["note"] = (f"{launches} launches each after a {warm_seconds:.0f}-s warm loop, random operands; "
                   "roofline.frac_normalised = in-situ dominant-kernel TFLOP/s / gemm_gate_up_tflops")
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cpu-full-step", action="store_true", help="only: one real full-depth B=1 oracle step on the host cores (minutes)")
    ap.add_argument("--no-kernel-pass", action="store_true", help="skip the untimed per-kernel HBM / attention event pass")
    ap.add_argument("--config", default="desta25_llama31-8B_Qformer6L")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--ctx", type=int, default=64)
    ap.add_argument("--tgt", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--encoder-overlap", action=argparse.BooleanOptionalAction, default=True,
                    help="next batch's frozen Whisper forward on its own HIP stream beside the LLM (default on; --no-encoder-overlap: at the end of the step on the main stream)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo for rehearsals)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--persistent-gemm", action="store_true", help="A/B: enable the persistent GEMM kernel")
    ap.add_argument("--no-stagger", action="store_true", help="A/B: lockstep GEMM schedule")
    ap.add_argument("--repeat", type=int, default=1, help="repeat the timed region (reports the best), for A/B runs")
    ap.add_argument("--no-dw-overlap", action="store_true", help="A/B: Q-Former weight gradients on the main stream")
    ap.add_argument("--gemm-4phase", action="store_true", help="A/B: the 4-phase (16 MFMAs per phase) GEMM schedule")
    ap.add_argument("--connector-overlap", action=argparse.BooleanOptionalAction, default=False,
                    help="the connector's backward joins all-reduce + Adafactor on the side stream (default off: with the encoder on its own stream it measured +0.5 ms)")
    ap.add_argument("--timed-gemm-events", action="store_true", help="A/B: HIP events around every GEMM launch INSIDE the timed region too (rounds 1-3 did; ~800 event records per step)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and run the gradient all-reduce even at WORLD_SIZE 1 (one-GPU rehearsal of the N > 1 path over RCCL)")
    ap.add_argument("--small-gemm-ring", type=int, default=None, help="A/B: option 6 of desta_gemm_set_option (0 = the 128x128 GEMM never takes its four-slot ring form, 1 = default, 2 = always)")
    ap.add_argument("--splitk-inkernel", action="store_true", help="A/B: reduce the K-slices of tail tiles inside the GEMM launch (scattered, ticketed) instead of by the fix-up launch")
    ap.add_argument("--launch-check", action="store_true", help="plumbing only: rendezvous + one all-reduce of the N ranks, no device work (CPU test of the self-launch path)")
    ap.add_argument("--no-calibration", action="store_true", help="skip the box_calibration block (fixed GEMM / copy workloads before the timed region)")
    ap.add_argument("--no-rope-fusion", action="store_true", help="A/B: rotary embedding as its own kernel (forward and backward) instead of inside the q|k|v GEMM epilogue / attention-backward stores")
    ap.add_argument("--attn-r2-backward", action="store_true", help="A/B: attention backward on round 2's kernels (separate delta, 4-wave dQ beside dK/dV); implies --no-rope-fusion")
    ap.add_argument("--attn-r2-forward", action="store_true", help="A/B: attention forward AND backward on round 2's 4-wave kernels; implies --no-rope-fusion")
    ap.add_argument("--lora", action="store_true", help="NOT the headline config: the reference's optional use_lora=True (rank-16 adapters on q/k/v of every decoder layer, full-row backward)")
    ap.add_argument("--no-dead-row-skip", action="store_true", help="A/B: the last decoder layer's o_proj / MLP (forward and backward) on every row instead of the target tail, layer 0's input gradient on every row instead of the audio rows")
    ap.add_argument("--attn-q64-two-kernels", action="store_true", help="A/B: the Q-Former's cross-attention backward on the separate dQ and dK/dV kernels instead of the one-pass kernel")
    ap.add_argument("--no-kv-side", action="store_true", help="A/B: the Q-Former's K | V projections inside the layer loop on the main stream instead of up front on a second stream")
    ap.add_argument("--data", choices=["synthetic", "wav"], default="synthetic",
                    help="wav: the REAL data path inside the timed region — B x 30-s RIFF/WAVE files (written under --wav-dir at start: half 16 kHz mono, "
                         "half 22.05 kHz stereo) -> BaseAudioTextDataset -> DataLoader workers (decode, resample, tokenise) -> H2D -> device log-mel; "
                         "NOT the headline line (BASELINE.json's metric is on synthetic inputs resident in HBM): printed beside it for VERDICT r3 item 8")
    ap.add_argument("--workers", type=int, default=4, help="--data wav: DataLoader worker processes (dataset.train_ds.num_workers); 0 = collate inline")
    ap.add_argument("--wav-dir", default="/tmp/desta_bench_wav")
    ap.add_argument("--main-priority", type=int, default=None, help="A/B: run the whole step on a non-default HIP stream of this priority (-1 = high) instead of torch's default stream")
    ap.add_argument("--side-priority", type=int, default=0, help="HIP priority of the encoder-prefetch / optimizer-tail streams (0 = same as the main stream: rounds 1-3; 1 = lower)")
    ap.add_argument("--no-gemm-tail-skip", action="store_true", help="A/B: the 256x256 GEMM re-loads dead LDS slots past its last K-tile (rounds 1-3) instead of stopping the half-tile stream there")
    ap.add_argument("--no-swiglu-fusion", action="store_true", help="A/B: silu(gate) * up and its backward as their own HBM passes instead of inside the gate|up / d(act) GEMM epilogues")
    ap.add_argument("--full-lm-head", action="store_true", help="A/B: lm_head / CE over the whole token grid, not only the target rows")
    a = ap.parse_args()

    if a.cpu_full_step:                                  # host-only leg, no device involved
        sys.path.insert(0, os.path.join(ROOT, "desta2.5-audio_amd"))
        from desta.models.modeling_desta25 import DeSTA25Config
        from desta.synthetic import FULL_CONFIGS
        print(json.dumps({"cpu_full_step": cpu_full_step(DeSTA25Config(**FULL_CONFIGS[a.config]), a.ctx, a.tgt,
                                                         min(len(os.sched_getaffinity(0)), 64)), "config": a.config}))
        return

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:    # not under torchrun: become the launcher (before any GPU call)
        sys.exit(self_launch(a, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if a.gpus != world:
        if rank == 0:
            print(f"[bench] --gpus {a.gpus} does not match WORLD_SIZE {world}: refusing to print a line for a different run",
                  file=sys.stderr)
        sys.exit(2)
    if a.launch_check:
        launch_check(a, world, rank)
        return
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.single_device:
        local = 0
    if a.force_dist:                                     # rehearsal of the N > 1 plumbing on ONE GPU: RCCL communicator, AVG all-reduce of
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")  # the gradient arena on the side stream, barrier fences, MAX over ranks
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ["DESTA_ALLREDUCE_WORLD1"] = "1"
    if world > 1 or a.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(a.backend)
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    from desta import _hip as H
    from desta.models.modeling_desta25 import DeSTA25AudioModel, DeSTA25Config
    from desta.synthetic import FULL_CONFIGS, RandomWeights, synthetic_inputs, synthetic_waveform
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments

    if a.persistent_gemm:
        H.gemm_set_option(0, 1)
    if a.no_stagger:
        H.gemm_set_option(1, 0)
    if a.gemm_4phase:
        H.gemm_set_option(4, 0)
    if a.small_gemm_ring is not None:
        H.gemm_set_option(6, a.small_gemm_ring)
    if a.splitk_inkernel:
        H.gemm_set_option(5, 1)
    if a.no_gemm_tail_skip:
        H.gemm_set_option(10, 0)
    cfg = DeSTA25Config(**FULL_CONFIGS[a.config], use_lora=a.lora)
    t0 = time.time()
    model = DeSTA25AudioModel(cfg, weights=RandomWeights(cfg, dev, seed=0), device=dev)
    args = TrainingArguments(learning_rate=1e-4, weight_decay=0.01, warmup_steps=5000, max_steps=10 ** 6, logging_steps=10 ** 9,
                             overlap_comm=not a.no_overlap, overlap_encoder=a.encoder_overlap and not a.no_overlap,
                             overlap_connector_backward=a.connector_overlap, side_stream_priority=a.side_priority)
    trainer = DeSTA25Trainer(model, args=args)
    if a.no_dw_overlap:
        model.connector.overlap_dw = False
    if a.full_lm_head:
        model.compact_lm_head = False
    if a.no_dead_row_skip:
        model.llm.skip_dead_rows = False
    if a.no_rope_fusion or a.attn_r2_backward or a.attn_r2_forward:
        model.llm.fuse_rope = False
    if a.attn_r2_backward:
        H.attention_set_option(3, 1)
    if a.attn_r2_forward:
        H.attention_set_option(0, 0)
    if a.no_swiglu_fusion:
        model.llm.fuse_swiglu = False
    if a.no_kv_side:
        model.connector.kv_side = False
    if a.attn_q64_two_kernels:
        H.attention_set_option(4, 0)
        model.connector.xattn_transposed = False

    B, S = a.batch, a.ctx + cfg.audio_tokens + a.tgt
    n_mels = cfg.encoder_config.num_mel_bins
    # two alternating synthetic batches per rank, resident in HBM (seed 1234 + rank)
    waves = [synthetic_waveform(B, dev, seed=1234 + rank + 97 * i) for i in range(2)]
    toks = [synthetic_inputs(cfg, B, a.ctx, a.tgt, dev, seed=1234 + rank + 97 * i) for i in range(2)]
    torch.cuda.synchronize()
    if rank == 0:
        print(f"[bench] model built in {time.time() - t0:.1f}s, trainable params {model.arena.true_numel() / 1e6:.2f} M, "
              f"HBM in use {torch.cuda.memory_allocated() / 2**30:.1f} GiB", file=sys.stderr)

    def batch(i):
        b = dict(toks[i % 2])
        b["batch_features"] = H.logmel(waves[i % 2], n_mels)              # A1 runs inside the step
        return b

    data_note = None
    if a.data == "wav":
        # the real data path: manifest records -> BaseAudioTextDataset -> per-epoch index batches -> DataLoader worker
        # processes running BaseCollateFn.host_collate -> BaseCollateFn.finish (H2D + device log-mel) in this process
        from desta.synthetic import WordTokenizer, write_synthetic_wav_dataset
        from desta.trainer.data.simple_dataset import BaseAudioTextDataset
        from desta.utils.audio import HipLogMelProcessor
        n_files = 4 * B
        recs = write_synthetic_wav_dataset(os.path.join(a.wav_dir, f"rank{rank}"), n_files, seed=1234 + rank)
        tokz = WordTokenizer(cfg.llm_config.vocab_size)
        tokz.pad_token, tokz.pad_token_id = tokz.eos_token, tokz.eos_token_id
        dcfg = {"model": {"audio_locator": cfg.audio_locator, "placeholder_token": "<|video_pad|>", "connector": {"prompt_size": cfg.prompt_size, "mode": "qformer_1"}}}
        # prompts / responses sized so that a sample is a.ctx context tokens + prompt_size audio tokens + a.tgt target tokens, like the synthetic batch
        probe = BaseAudioTextDataset(dcfg, {"data_root": os.path.join(a.wav_dir, f"rank{rank}"), "max_seq_length": S}, tokz, None, records=recs[:1])
        n_ctx0 = len(tokz.tokenize(probe[0]["audio_context"])) - cfg.prompt_size
        for r_ in recs:
            r_["prompt"] = r_["prompt"] + "".join(f" w{j}" for j in range(max(0, a.ctx - n_ctx0)))
            r_["response"] = " ".join(f"t{j % 97}" for j in range(a.tgt))
        wav_ds = BaseAudioTextDataset(dcfg, {"data_root": os.path.join(a.wav_dir, f"rank{rank}"), "max_seq_length": S}, tokz,
                                      HipLogMelProcessor(n_mels, dev), records=recs)
        trainer.train_dataset, trainer.data_collator = wav_ds, wav_ds.collate_fn
        trainer.args.per_device_train_batch_size, trainer.args.dataloader_num_workers = B, a.workers
        wav_iter = {"it": None, "epoch": 0}

        def batch(i):                                                     # noqa: F811 — the next collated batch of the (cycled) epoch stream
            while True:
                if wav_iter["it"] is None:
                    wav_iter["it"] = iter(trainer._epoch_batches(wav_iter["epoch"]))
                    wav_iter["epoch"] += 1
                b = next(wav_iter["it"], None)
                if b is not None:
                    return b
                wav_iter["it"] = None
        b0 = batch(0)
        data_note = (f"{n_files} RIFF/WAVE clips of 30 s per rank (half 16 kHz mono, half 22.05 kHz stereo PCM16) -> BaseAudioTextDataset -> "
                     f"DataLoader(num_workers={a.workers}, pin_memory, prefetch 2) running BaseCollateFn.host_collate -> finish (H2D + device log-mel); "
                     f"batch {tuple(b0['input_ids'].shape)} tokens, features {tuple(b0['batch_features'].shape)}")
        assert tuple(b0["input_ids"].shape) == (B, S), (b0["input_ids"].shape, (B, S))

    step_events = []
    carried = {}                                         # {index: batch} prefetched by the previous run()'s last step

    def run(nsteps, start, mark=False):
        # one continuous training loop cut into warm-up / timed / kernel-pass segments: the batch (and its frozen-encoder
        # prefetch) that the last step of a segment prepared is the first batch of the next one, so a segment of K steps holds
        # exactly K log-mel + Whisper forwards (for batch t+1 inside step t), as K steady-state steps of a real run do
        cur = carried.pop(start, None)
        carried.clear()
        if cur is None:
            cur = batch(start)
        loss = None
        if mark:
            step_events.append(torch.cuda.Event(enable_timing=True))
            step_events[-1].record()
        for i in range(start, start + nsteps):
            nxt = batch(i + 1)
            loss = trainer.training_step(cur, nxt)
            cur = nxt
            if mark:                                     # main-stream step boundary (no sync): per-step spread for mean +- sd
                step_events.append(torch.cuda.Event(enable_timing=True))
                step_events[-1].record()
        carried[start + nsteps] = cur
        trainer.wait_update()
        return loss

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        print(f"[bench] HIP stream priority range (least, greatest): {torch.cuda.Stream.priority_range()}; side streams at {a.side_priority}"
              + (f", main stream at {a.main_priority}" if a.main_priority is not None else ""), file=sys.stderr)
    if a.main_priority is not None:
        _main = torch.cuda.Stream(device=dev, priority=a.main_priority)
        _main.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(_main)
    calib = None
    if rank == 0 and not a.no_calibration:
        calib = box_calibration(H, dev)
    fence()
    run(a.warmup, 0)
    fence()
    trainer.comm_profile = [] if (world > 1 or a.force_dist) else None     # HIP events around the collective / the main stream's wait
    if a.timed_gemm_events:
        H.gemm_profile_start()
    t0 = time.perf_counter()
    loss = run(a.steps, a.warmup, mark=True)
    fence()
    elapsed = time.perf_counter() - t0
    prof_timed = H.gemm_profile_stop(by_kernel=True) if a.timed_gemm_events else None
    comm, trainer.comm_profile = trainer.comm_profile, None
    per_step = [x.elapsed_time(y) for x, y in zip(step_events[:-1], step_events[1:])]
    # UNTIMED, UN-OVERLAPPED pass (the `roofline` leg): the same steps with EVERY side stream off — optimizer tail, connector backward,
    # encoder prefetch, K | V projections and weight gradients all in line on the main stream — and HIP events around every GEMM
    # launch and every tagged HBM-bound / attention launch.  Beside another stream two kernels time-share the chip and each one's
    # duration reads long for a step that is SHORTER; the timed region above keeps every overlap (steps/s is the metric), the
    # per-kernel rates come from here (a kernel alone on the device, in situ: same operands, same cache state as in the step).
    prof, kprof, kp_steps = {}, {}, 3
    if not a.no_kernel_pass:                             # every rank steps (the all-reduce is collective); rank 0 records
        saved = (trainer._side, trainer._enc_stream, model.connector.kv_side, model.connector.overlap_dw)
        trainer.wait_update()
        torch.cuda.synchronize()
        trainer._side, trainer._enc_stream, model.connector.kv_side, model.connector.overlap_dw = None, None, False, False
        run(1, a.warmup + a.steps)                       # one un-recorded step: the overlapped pipeline's in-flight prefetch drains
        if rank == 0:
            H.gemm_profile_start()
            H.kernel_profile_start()
        run(kp_steps, a.warmup + a.steps + 1)
        if rank == 0:
            kprof = H.kernel_profile_stop()
            prof = H.gemm_profile_stop(by_kernel=True)
        trainer._side, trainer._enc_stream, model.connector.kv_side, model.connector.overlap_dw = saved
    fence()
    n_launch, flops, gemm_ms = prof.get(2, (0, 0.0, 0.0))                 # the dominant kernel: gemm_bf16_nt_256_kernel
    n_other = sum(v[0] for k, v in prof.items() if k != 2)
    ms_other = sum(v[2] for k, v in prof.items() if k != 2)
    flops_other = sum(v[1] for k, v in prof.items() if k != 2)
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss)

    if rank == 0:
        achieved = flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        # L2-boundary bytes per launch of the dominant kernel: from separate rocprofv3 --pmc passes of this same command, committed
        # under profiles/ (FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes) — a STORED record, not a counter of
        # this run (hardware counters need the profiler around the process); null if absent
        traffic, traffic_src = None, None
        tpath = next((pth for pth in (os.path.join(ROOT, "profiles", f) for f in ("r04_gemm_hbm_traffic.json", "r03_gemm_hbm_traffic.json",
                                                                                    "r02_gemm_hbm_traffic.json")) if os.path.isfile(pth)), "")
        if a.config == "desta25_llama31-8B_Qformer6L" and os.path.isfile(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj["fetch_bytes_per_launch"] + tj["write_bytes_per_launch"]
            traffic_src = {"measured_in_this_run": False, "file": os.path.relpath(tpath, ROOT),
                           "what": "TCC FETCH_SIZE x 2 + WRITE_SIZE per launch (L2 boundary: Infinity-Cache hits included), separate --pmc passes",
                           "hbm_side": {k: tj[k] for k in ("dram_read_bytes_per_launch", "dram_read_counter", "hbm_read_fraction_of_fetch") if k in tj} or None}
        ms_step = 1e3 * elapsed / a.steps
        # reference point measured on the same hardware with tools/hf_step_bench.py (the step composed from stock PyTorch-ROCm /
        # transformers modules, as the reference composes it); NOT `vs_baseline` (BASELINE.md publishes no number for this metric)
        torch_ref = None
        rpath = os.path.join(ROOT, "profiles", "r04_hf_pytorch_step.json")
        if not os.path.isfile(rpath):
            rpath = os.path.join(ROOT, "profiles", "r02_hf_pytorch_step.json")
        if a.config == "desta25_llama31-8B_Qformer6L" and world == 1 and os.path.isfile(rpath):
            with open(rpath) as f:
                torch_ref = json.load(f)
            torch_ref["speedup_of_this_run"] = torch_ref["ms_per_step"] / ms_step
            torch_ref["measured_in_this_run"] = False
            torch_ref["note"] = (f"stored constant: tools/hf_step_bench.py measured once ({torch_ref.get('log')}; round 4: in the same lease as "
                                 "profiles/r04_bench_driver_cmd.log, a box of gate_up calibration 1355 TFLOP/s where this bench ran 154.7 ms); only the "
                                 "ratio uses this run's ms_per_step")
        mean_ps = sum(per_step) / max(1, len(per_step))
        sd_ps = (sum((x - mean_ps) ** 2 for x in per_step) / max(1, len(per_step) - 1)) ** 0.5
        # executed FLOP per step: every GEMM launch of a step (HIP-event records carry 2MNK) + the attention kernels'
        # MFMA FLOP (4 Sq Sk D per head forward, x2.5 backward, halved under the causal mask), both from the un-overlapped pass
        attn = {t: v for t, v in kprof.items() if t.startswith("attn_")}
        attn_flop_step = sum(v[1] for v in attn.values()) / kp_steps
        gemm_flop_step = (flops + flops_other) / kp_steps
        exec_flop = gemm_flop_step + attn_flop_step
        hbm_kernels = {t: {"calls_per_step": v[0] / kp_steps, "algorithmic_GB_per_step": v[1] / kp_steps / 1e9, "ms_per_step": v[2] / kp_steps,
                           "GBps": v[1] / (v[2] * 1e-3) / 1e9 if v[2] > 0 else 0.0, "frac_of_hbm_peak": (v[1] / (v[2] * 1e-3) / 1e9 / HBM_PEAK_GBS) if v[2] > 0 else 0.0}
                       for t, v in sorted(kprof.items()) if not t.startswith("attn_")}
        attn_kernels = {t: {"calls_per_step": v[0] / kp_steps, "GFLOP_per_call": v[1] / max(v[0], 1) / 1e9, "us_per_call": 1e3 * v[2] / max(v[0], 1),
                            "TFLOPs": v[1] / (v[2] * 1e-3) / 1e12 if v[2] > 0 else 0.0,
                            "frac_of_mfma_peak": (v[1] / (v[2] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS) if v[2] > 0 else 0.0}
                        for t, v in sorted(attn.items())}
        overlaps = []
        if not a.no_overlap:
            overlaps.append("all-reduce + clip + Adafactor + weight re-cast of step t on a side stream")
            if a.connector_overlap:
                overlaps.append("the connector's backward of step t on that side stream too")
        if a.encoder_overlap:
            overlaps.append("the frozen Whisper forward of batch t+1 on its own HIP stream beside the connector / LLM of batch t")
        elif not a.no_overlap:
            overlaps.append("the frozen Whisper forward of batch t+1 on the main stream at the end of step t, beside the side-stream tail")
        out = {
            "metric": "train steps/sec (node) Whisper-v3+Llama3.1-8B Q-Former6L at 1/2/4/8 MI355X",
            # OPTIMIZER steps per second of the node; one step = one update with per-GPU batch B, global batch B x N
            "value": a.steps / elapsed, "unit": "steps/s",
            "samples_per_s": world * B * a.steps / elapsed,
            "gpu_batch_steps_per_s": world * a.steps / elapsed,           # per-GPU batch-steps summed over the node (what rounds 1-3 printed as `value`)
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "ms_per_step_mean_sd": [mean_ps, sd_ps], "ms_per_step_min_max": [min(per_step), max(per_step)] if per_step else None,
            "ms_per_step_each": [round(x, 2) for x in per_step],        # main-stream event deltas, in order
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic" if a.data == "synthetic" else f"synthetic WAVE files through the real data path: {data_note}",
            "config": {"workload": f"{a.config}: {os.path.basename(cfg.encoder_model_id)} + {os.path.basename(cfg.llm_model_id)}, "
                                   f"Q-Former {cfg.qformer_num_hidden_layers}L, per-GPU batch {B} x 30 s clips, "
                                   f"S={S} ({a.ctx} ctx + {cfg.audio_tokens} audio + {a.tgt} target tokens), random-init weights at true shapes"
                                   + (" + ORCA hybrid (gated cross-attention behind every decoder layer trainable: NOT the headline config)" if cfg.connector_mode == "orca_hybrid" else "")
                                   + (" + use_lora (rank-16 q/k/v adapters trainable: NOT the headline config)" if a.lora else ""),
                       "global_batch": B * world, "seq_len": S, "parallelism": f"dp{world}",
                       "step_definition": "one optimizer step of the node: every GPU runs the hot path over its own batch of "
                                          f"{B} clips (global batch {B * world}), one mean all-reduce of the gradient arena, one update; "
                                          "value = K / max-over-ranks time (weak scaling: ideal is a constant value as N grows; "
                                          "samples_per_s = value x global batch is the throughput that grows with N)",
                       "stream_overlap": "; ".join(overlaps) if overlaps else "none",
                       "training_fast_path": ("off (--full-lm-head): lm_head / CE over the whole token grid, backward over every row"
                                              if a.full_lm_head else
                                              "on: lm_head / CE on the rows that carry a target, LLM backward from the first audio span "
                                              "(rows whose logits the loss ignores / whose gradient nothing consumes are not computed; "
                                              "loss and every parameter gradient identical to the full grid, tests/test_gpu_model.py)")},
            "final_loss": final_loss,
            "box_calibration": calib,
            "pytorch_rocm_reference_point": torch_ref,
            "roofline": {"bound": "mfma", "kernel": "gemm_bf16_nt_256_kernel", "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "measured_in": (f"untimed pass of {kp_steps} steps right after the timed region with every side stream OFF (a kernel's HIP-event "
                                         "duration beside another stream includes the time-sharing); the timed region runs WITH the overlaps"),
                         # in-situ rate of the dominant kernel / the same kernel alone on this box right before the run (gate_up shape)
                         "frac_normalised": (achieved / calib["gemm_gate_up_tflops"]) if calib and achieved > 0 else None,
                         "launches_per_step": n_launch / kp_steps, "avg_launch_us": 1e3 * gemm_ms / max(n_launch, 1),
                         "flop_per_launch": flops / max(n_launch, 1), "gemm_ms_per_step": gemm_ms / kp_steps,
                         "other_gemm_kernels": {"launches_per_step": n_other / kp_steps, "ms_per_step": ms_other / kp_steps,
                                                "tflops": (flops_other / (ms_other * 1e-3) / 1e12) if ms_other > 0 else 0.0,
                                                "note": "gemm_bf16_nt_kernel family (128x128 and the connector's small-GEMM kernels, incl. transposed-storage dW)"},
                         # the WHOLE step against the same peak: executed FLOP (GEMMs + attention MFMA work; the fast path skips
                         # lm_head rows without a target and the backward in front of the first audio span) / wall time of the TIMED region
                         "whole_step": {"executed_flop_per_step": exec_flop, "gemm_flop_per_step": gemm_flop_step,
                                        "attention_flop_per_step": attn_flop_step, "achieved": exec_flop / (ms_step * 1e-3) / 1e12,
                                        "frac": exec_flop / (ms_step * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                        "reference_flop_per_step": 1.82e14,
                                        "frac_on_reference_flop": 1.82e14 / (ms_step * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS
                                        if a.config.startswith("desta25_llama31-8B") and (B, a.ctx, a.tgt) == (8, 64, 512) else None},
                         # HBM-bound kernels of the step: ALGORITHMIC bytes / HIP-event time of the same untimed pass, vs 8 TB/s
                         "hbm_kernels": hbm_kernels, "attention_kernels": attn_kernels},
        }
        if prof_timed is not None:                                        # --timed-gemm-events: the rounds-1-3 bookkeeping, for comparison
            n2, f2, m2 = prof_timed.get(2, (0, 0.0, 0.0))
            out["roofline"]["in_timed_region_with_overlaps"] = {"achieved": f2 / (m2 * 1e-3) / 1e12 if m2 > 0 else 0.0, "gemm_ms_per_step": m2 / a.steps,
                                                                 "launches_per_step": n2 / a.steps}
        if comm is not None:
            # data-parallel exchange of the timed region (rank 0's events): the flat fp32 gradient arena, one collective per step
            ar = [x.elapsed_time(y) for kind, x, y in comm if kind == "allreduce"]
            st = [x.elapsed_time(y) for kind, x, y in comm if kind == "wait_update"]
            out["allreduce_ms"] = sum(ar) / max(1, len(ar))
            out["allreduce_ms_min_max"] = [min(ar), max(ar)] if ar else None
            out["bytes_allreduced"] = int(model.arena.grads.numel() * 4)
            out["allreduce_busbw_GBps"] = (2 * (world - 1) / world * out["bytes_allreduced"] / (out["allreduce_ms"] * 1e-3) / 1e9) if ar and world > 1 and out["allreduce_ms"] > 0 else None
            out["wait_update_stall_ms"] = sum(st) / max(1, len(st))        # main stream idle in front of the connector forward, waiting for the side-stream tail (connector backward + all-reduce + Adafactor + re-cast) of the previous step
            out["comm_hidden"] = bool(st) and out["wait_update_stall_ms"] < 0.05
            out["comm_backend"] = dist.get_backend() if dist.is_initialized() else None
        if cfg.connector_mode == "orca_hybrid" and not a.no_cpu_baseline:
            out["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": 0, "kind": "port",
                                   "sample": "not timed for the ORCA-hybrid side configs (the host leg restates the qformer_1 headline step)"}
        elif not a.no_cpu_baseline and world == 1:
            del trainer, model                                            # the host legs below need none of it
            torch.cuda.empty_cache()
            try:
                threads = min(len(os.sched_getaffinity(0)), 64)
                out["cpu_baseline"] = cpu_baseline(cfg, B, a.ctx, a.tgt, threads)
                out["cpu_baseline"]["debug_config"] = cpu_baseline_debug(threads)
                if a.steps >= 20 and a.config == "desta25_llama31-8B_Qformer6L" and not a.lora:
                    # ONE real full-depth step of the oracle at true shapes (B = 1, fp32) in THIS run: `value` is taken from it
                    # (x B samples per step), the one-layer extrapolation stays beside it as a cross-check
                    full = cpu_full_step(cfg, a.ctx, a.tgt, threads, steps=1)
                    cb = out["cpu_baseline"]
                    cb["value_extrapolated_from_one_layer"] = cb["value"]
                    cb["value"] = 1.0 / (B * full["seconds_per_sample_step"])
                    cb["full_step"] = full
                    cb["sample"] = (f"ONE real full-depth oracle step in this run: B=1, fp32, {threads} threads, "
                                    f"{full['seconds_per_sample_step']:.1f} s (32 Whisper + 4x6 Q-Former + 32 LLM layers, lm_head, CE, backward, clip, "
                                    f"Adafactor); value = 1 / (B={B} x that).  Cross-check, one layer per stack scaled up: " + cb["sample"])
            except Exception as ex:                                   # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": 0, "kind": "port", "sample": f"failed: {ex}"}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
