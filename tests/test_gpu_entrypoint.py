"""GPU: the entry point end to end — `main([...])` trains the debug config, writes HF `checkpoint-<step>/` directories at
epoch ends, and `resume_from_checkpoint=` (handed to trainer.train like examples/train/train_desta.py:231 does) continues an
interrupted run bit for bit: parameters, Adafactor moments, schedule position, step and dropout stream."""
import importlib.util
import json
import os

import pytest
import torch
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mod():
    spec = importlib.util.spec_from_file_location("train_desta", os.path.join(ROOT, "examples", "train", "train_desta.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_interrupted_6_plus_6_equals_12_steps_through_main(tmp_path):
    m = _mod()
    common = ["--config-name", "desta25_debug", "+dataset=debug", "trainer.max_steps=-1", "trainer.max_epochs=2",
              "dataset.train_ds.num_samples=12", "optim.sched.warmup_steps=3", "optim.lr=1e-3"]       # 6 steps per epoch, 12 in all
    a = m.main(common + [f"exp_dir={tmp_path}/a"])
    assert a.global_step == 12 and a.total_steps == 12
    for step in (6, 12):
        d = tmp_path / "a" / f"checkpoint-{step}"
        assert sorted(os.listdir(d)) == ["config.json", "desta_hip_state.json", "model.safetensors", "optimizer.pt", "rng_state.pth", "scheduler.pt",
                                         "trainer_state.json", "training_args.bin"]
    assert os.path.isdir(tmp_path / "a" / "checkpoint-initial")
    # optimizer.pt is the transformers.Adafactor wire format (two groups, factored moments)
    sd = torch.load(tmp_path / "a" / "checkpoint-6" / "optimizer.pt", weights_only=True)
    assert len(sd["param_groups"]) == 2 and sd["state"][0]["step"] == 6 and "exp_avg_sq_row" in sd["state"][0]
    sched = torch.load(tmp_path / "a" / "checkpoint-6" / "scheduler.pt", weights_only=True)
    assert sched["last_epoch"] == 6
    # LR decays linearly to zero over the 12 steps derived from epochs x steps-per-epoch (was constant)
    assert a.get_last_lr() == 0.0
    # the "interrupted" run: a fresh process state resumed from the step-6 checkpoint of run A
    b = m.main(common + [f"exp_dir={tmp_path}/b", f"resume_from_checkpoint={tmp_path}/a/checkpoint-6"])
    assert b.global_step == 12
    assert not os.path.isdir(tmp_path / "b" / "checkpoint-initial")              # reference: only when not resuming
    pa, pb = load_file(tmp_path / "a" / "checkpoint-12" / "model.safetensors"), load_file(tmp_path / "b" / "checkpoint-12" / "model.safetensors")
    assert pa.keys() == pb.keys()
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k
    oa = torch.load(tmp_path / "a" / "checkpoint-12" / "optimizer.pt", weights_only=True)
    ob = torch.load(tmp_path / "b" / "checkpoint-12" / "optimizer.pt", weights_only=True)
    for i in oa["state"]:
        for k, v in oa["state"][i].items():
            assert (torch.equal(v, ob["state"][i][k]) if torch.is_tensor(v) else v == ob["state"][i][k]), (i, k)
    with open(tmp_path / "b" / "checkpoint-12" / "trainer_state.json") as f:
        assert json.load(f)["global_step"] == 12
    # the parameters did move between the checkpoints (the comparison above is not vacuous)
    p6 = load_file(tmp_path / "a" / "checkpoint-6" / "model.safetensors")
    assert any(not torch.equal(p6[k], pa[k]) for k in pa)
    with pytest.raises(AssertionError, match="Cannot provide both"):
        m.main(common + [f"exp_dir={tmp_path}/c", f"resume_from_checkpoint={tmp_path}/a/checkpoint-6", "init_from_pretrained_weights=/x.ckpt"])


def test_manifest_dataset_through_main(tmp_path, monkeypatch):
    """Non-synthetic datasets: JSONL manifest + WAVE files -> `BaseAudioTextDataset` (disk cache under HF_HOME) -> seeded
    DistributedSampler order -> `BaseCollateFn` (device log-mel) -> the trainer, through `main([...])` like the reference's
    entry point builds them (train_desta.py:196-214).  The tokenizer is injected (no hub access)."""
    import sys
    import wave

    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import ToyTokenizer
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf"))
    rng = np.random.default_rng(0)
    audio = tmp_path / "audio"
    audio.mkdir()
    recs = []
    for i in range(5):
        x = 0.1 * rng.standard_normal(16000 + 4000 * i)
        with wave.open(str(audio / f"clip{i}.wav"), "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())
        recs.append(dict(id=f"clip{i}.wav", prompt=f"describe clip number {i}", response="some noise " * (1 + i % 3)))
    recs.append(dict(id="clip0.wav", prompt="", response="skipped: empty prompt"))
    man = tmp_path / "train.jsonl"
    man.write_text("".join(json.dumps(r) + "\n" for r in recs))
    m = _mod()
    monkeypatch.setattr(m, "create_tokenizer", lambda cfg: ToyTokenizer(vocab_size=512))
    tr = m.main(["--config-name", "desta25_debug", "+dataset=debug", f"exp_dir={tmp_path}/exp", "trainer.max_epochs=2", "trainer.max_steps=-1",
                 "optim.sched.warmup_steps=0", "optim.lr=1e-3", "dataset.train_ds.synthetic=false", f"dataset.train_ds.data_root={audio}",
                 f"dataset.train_ds.manifest_filepaths=[{man}]", "dataset.validation_ds.synthetic=false",
                 f"dataset.validation_ds.manifest_filepaths=[{man}]", f"dataset.validation_ds.data_root={audio}",
                 "model.generation_kwargs.max_new_tokens=3"])
    assert len(tr.train_dataset) == 5                                          # the empty-prompt record is filtered
    assert tr.steps_per_epoch() == 3 and tr.total_steps == 6 and tr.global_step == 6          # ceil(5 / 2) steps x 2 epochs
    assert os.path.isdir(tmp_path / "exp" / "checkpoint-3") and os.path.isdir(tmp_path / "exp" / "checkpoint-6")
    assert os.path.exists(str(tr.train_dataset.cache_paths([str(man)])[2]))    # the `.ready` file of the disk cache
    losses = [h["train/loss"] if "train/loss" in h else h.get("loss") for h in tr.log_history if ("train/loss" in h or "loss" in h)]
    assert all(l is None or (l == l and l > 0) for l in losses)
    # the reference's validation loop (desta_trainer.py:104-158): loss / ppl, generation, predictions JSONL + accuracy report
    val = tmp_path / "exp" / "results" / "val"
    # the entry point evaluated BEFORE training (reference :220-224) and, with `val_check_interval: 1.0` (a float -> eval_strategy
    # "epoch", :147), at both epoch ends: three prediction files, the first at step 0
    assert sorted(os.listdir(val / "preds")) == ["val@ep=0.0-0.jsonl", "val@ep=1.0-3.jsonl", "val@ep=2.0-6.jsonl"]
    evals = [h for h in tr.log_history if "eval_loss" in h]
    assert len(evals) == 3 and all(h["eval_loss"] > 0 for h in evals)
    ev = tr.evaluate(generation_kwargs={"max_new_tokens": 3})
    assert {"eval_loss", "eval_ppl", "eval_accuracy", "eval_accuracy_by_category", "eval_acc/all"} <= set(ev) and ev["eval_loss"] > 0
    assert len(tr.prediction_step_outputs) == 5 and 0.0 <= ev["eval_accuracy"] <= 1.0
    assert len(os.listdir(val / "preds")) == 4 and any(f.endswith("-report.json") for f in os.listdir(val))     # same step again: a new file, nothing overwritten
    rows = [json.loads(l) for l in open(val / "preds" / "val@ep=2.0-6-1.jsonl")]
    assert len(rows) == 5 and all({"prediction", "label", "context", "correct", "index"} <= set(r) for r in rows)
    # epoch order: a permutation from seed + epoch, every sample once per epoch
    seen = []
    tr.data_collator = lambda rows: [r["id"] if "id" in r else r["processed_audios"][0]["audio"] for r in rows]
    for ep in (0, 1):
        seen.append([x for b in tr._epoch_batches(ep) for x in b])
    assert sorted(seen[0]) == sorted(seen[1]) and len(set(seen[0])) == 5 and seen[0] != seen[1]


def test_init_from_pretrained_weights_lightning_style_file(tmp_path):
    """`init_from_pretrained_weights=<file>` (reference examples/train/train_desta.py:73-83, :139-145): a legacy Lightning
    checkpoint `{"state_dict": {"model.<name>": tensor}}` — the "model." prefix is stripped, names the model does not train
    (frozen LLM / Whisper tensors that such files also hold) are ignored (`strict=False`), tensors absent from the file keep
    their initial value.  The file is hand-made here (tensors only: the product loads it with weights_only=True)."""
    m = _mod()
    args = ["--config-name", "desta25_debug", "+dataset=debug", "trainer.max_steps=1", f"exp_dir={tmp_path}/base"]
    base = m.main(args).model
    names = base.trainable_parameter_names
    p0 = {n: base.arena.param(n).detach().cpu().clone() for n in names}
    del base
    g = torch.Generator().manual_seed(3)
    picked = [n for n in names if "layer.0.crossattention.self.query" in n or n.endswith("layer_weights") or ".proj.1.bias" in n]
    assert len(picked) == 4
    new = {n: torch.randn(p0[n].shape, generator=g) for n in picked}
    blob = {"state_dict": {**{"model." + n: v for n, v in new.items()},
                           "model.llm_model.model.embed_tokens.weight": torch.zeros(4, 4),         # frozen tensor in a Lightning file: ignored
                           "epoch_marker_not_a_parameter": torch.zeros(1)},
            "epoch": 3, "global_step": 1234}
    path = str(tmp_path / "legacy.ckpt")
    torch.save(blob, path)
    tr = m.main(["--config-name", "desta25_debug", "+dataset=debug", "trainer.max_steps=1", f"exp_dir={tmp_path}/init",
                 f"init_from_pretrained_weights={path}"])
    init = load_file(tmp_path / "init" / "checkpoint-initial" / "model.safetensors")              # written right after the load, before any step
    for n in picked:
        assert torch.equal(init[n], new[n]), n
    # untouched tensors equal a model built without the file (same seeded init), touched ones equal the file
    fresh = m.main(["--config-name", "desta25_debug", "+dataset=debug", "trainer.max_steps=1", f"exp_dir={tmp_path}/fresh"])
    f0 = load_file(tmp_path / "fresh" / "checkpoint-initial" / "model.safetensors")
    for n in names:
        if n in new:
            assert not torch.equal(init[n], f0[n]), n
        else:
            assert torch.equal(init[n], f0[n]), n
    assert tr.global_step == 1                                               # and the run trains on from the loaded weights
    # a plain state dict without the "state_dict" wrapper loads too
    torch.save({"model." + n: v for n, v in new.items()}, str(tmp_path / "flat.ckpt"))
    m.load_pretrained_weights(fresh.model, str(tmp_path / "flat.ckpt"))
    for n in picked:
        assert torch.equal(fresh.model.arena.param(n).detach().cpu(), new[n]), n


def test_orca_hybrid_through_main(tmp_path):
    """`model.connector.mode=orca_hybrid` + the `model.orca.*` keys of the reference's ORCA YAMLs through the entry point: the dataset
    sizes every audio span by `global_num_tokens` placeholders (simple_dataset.py: `orca_global_num_tokens` in ORCA mode), the trainer
    optimises LM loss + the ORCA losses, the checkpoint holds the ORCA state-dict keys, and resuming continues bit for bit."""
    m = _mod()
    orca = ["model.connector.mode=orca_hybrid", "+model.orca.enabled=true", "+model.orca.local_enabled=true", "+model.orca.global_cross_attn=true",
            "+model.orca.deep_injection_enabled=true", "+model.orca.global_num_tokens=8", "+model.orca.local_downsample=4", "+model.orca.local_kernel_size=5",
            "+model.orca.gate_init=0.1", "+model.orca.audio_position_scale=2.5", "+model.orca.ortho_weight_global=0.05",
            "+model.orca.ortho_diversity_weight=0.05", "+model.orca.ortho_weight_qformer_local=0.05", "+model.orca.align_weight_local=0.05"]
    common = ["--config-name", "desta25_debug", "+dataset=debug", "trainer.max_steps=-1", "trainer.max_epochs=2", "dataset.train_ds.num_samples=8",
              "optim.sched.warmup_steps=2", "optim.lr=1e-3"] + orca
    a = m.main(common + [f"exp_dir={tmp_path}/a"])
    assert a.global_step == 8 and a.model.config.connector_mode == "orca_hybrid" and a.model.orca is not None
    assert a.model.config.orca_global_num_tokens == 8 and a.model.config.orca_global_cross_attn and a.model.config.audio_tokens == 8
    sd = load_file(tmp_path / "a" / "checkpoint-8" / "model.safetensors")
    assert any(k.startswith("orca_cross_attns.1.cross_attn.") for k in sd) and "perception.connector.local_conv.weight" in sd
    assert "perception.connector.global_queries.3" in sd and not any("layer_prompts" in k for k in sd)
    b = m.main(common + [f"exp_dir={tmp_path}/b", f"resume_from_checkpoint={tmp_path}/a/checkpoint-4"])
    pb = load_file(tmp_path / "b" / "checkpoint-8" / "model.safetensors")
    assert sd.keys() == pb.keys() and all(torch.equal(sd[k], pb[k]) for k in sd)
    p4 = load_file(tmp_path / "a" / "checkpoint-4" / "model.safetensors")
    assert any(not torch.equal(p4[k], sd[k]) for k in sd)
