#!/usr/bin/env python
"""Per-TENSOR gradient error of the reference's own precision policy.

For every golden made by the reference's fp32 forward + autograd (ref_tiny_llama, ref_deep_llama, ref_deep_qwen3,
ref_tied_qwen3) the oracle is run under `O.autocast_bf16()` — the hand restatement of torch.autocast("cuda", bf16) + bf16 LLM
that the reference trains under (hazard H11) — and the relative L2 error of each connector gradient against the fp32 golden is
written to tests/golden/autocast_policy_grad_errors.json.  That file is the yardstick the GPU test holds the HIP path's
per-tensor errors against (tests/test_gpu_model.py): a tensor where bf16 arithmetic ITSELF loses 10 % (small-norm
cross-attention query weights at depth) is the policy's floor, a tensor where only the HIP path does is a rounding bug.

    python tests/golden/make_policy_grad_errors.py        # CPU, ~1 min; needs only the committed goldens + oracle/
"""
import json
import os
import sys

import torch
from safetensors.torch import load_file

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import desta_oracle as O  # noqa: E402

CASES = {"ref_tiny_llama": lambda: O.tiny_dims(False), "ref_tiny_qwen3": lambda: O.tiny_dims(True), "ref_deep_llama": lambda: O.deep_dims(False),
         "ref_deep_qwen3": lambda: O.deep_dims(True), "ref_tied_qwen3": O.tied_dims}


def main():
    out = {}
    for name, mk in CASES.items():
        g = load_file(os.path.join(HERE, name + ".safetensors"))
        d = mk()
        w = O.init_weights(d, seed=7)
        n_a = g["starts"].shape[0]
        batch = {"input_ids": g["input_ids"], "attention_mask": g["attention_mask"], "labels": g["labels"],
                 "batch_features": g["batch_features"], "batch_start_positions": [(int(b), int(s)) for b, s in g["starts"].tolist()],
                 "batch_transcription_ids": [torch.zeros(1, 0, dtype=torch.long) for _ in range(n_a)]}
        names = O.trainable_names(d)
        for n in names:
            w[n].requires_grad_(True)
        with O.autocast_bf16():
            loss, _ = O.model_forward(w, d, batch)
        loss.backward()
        gn = sorted(float(g["grad::" + n].double().norm()) for n in names)
        floor = gn[len(gn) // 2] * 1e-2                   # same floor as tests/test_gpu_model.py (numerically-zero gradients: key biases)
        errs = {}
        for n in names:
            ref = g["grad::" + n].double()
            errs[n] = float((w[n].grad.double() - ref).norm() / max(float(ref.norm()), floor))
        worst = sorted(errs, key=errs.get)[-5:]
        print(name, [(n.split("connector.")[-1], round(errs[n], 4)) for n in reversed(worst)])
        out[name] = {n: round(e, 6) for n, e in errs.items()}
    with open(os.path.join(HERE, "autocast_policy_grad_errors.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
