#!/usr/bin/env python
"""Entry point of the MI355X-native DeSTA2.5 training path.

Same command line and YAML schema as the reference's examples/train/train_desta.py (Hydra style):

    torchrun --nproc_per_node=N examples/train/train_desta.py --config-name desta25_llama31-8B_Qformer6L \\
        +dataset=synthetic exp_dir=/tmp/exp [key.sub=value ...]

Hydra/OmegaConf are not installed here, so the few features the reference uses are restated on plain
`yaml.safe_load`: `--config-name`, the `+dataset=<group file>` addition, dotted `key=value` overrides and
the `???` mandatory marker.  `create_model` / `create_training_args` read exactly the keys the reference
reads (train_desta.py:96-162).  Checkpoints are HF `checkpoint-<step>/` directories written at every epoch end
(save_strategy="epoch", :146) and `resume_from_checkpoint` is handed to `trainer.train` (:231), which restores the
parameters, Adafactor moments, schedule position and step.  Datasets: `synthetic: true` streams (benchmarks, tests), or
the reference's JSONL manifests through `desta.trainer.data.simple_dataset` (`BaseAudioTextDataset` + `BaseCollateFn`, WAV
decode + device log-mel) — that path needs the LLM's tokenizer, which is loaded by NAME from the local HF cache
(`create_tokenizer`; there is no hub access here, tests inject one).
"""
import argparse
import logging
import os
import sys

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "desta2.5-audio_amd"))


class Cfg(dict):
    """dict with attribute access (the OmegaConf surface the entry point needs)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return Cfg(v) if isinstance(v, dict) and not isinstance(v, Cfg) else v

    def get(self, k, default=None):
        v = dict.get(self, k, default)
        return Cfg(v) if isinstance(v, dict) and not isinstance(v, Cfg) else v


_NUM = __import__("re").compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)$")


def _numify(x):
    """YAML 1.1 reads `1e-4` as a string; OmegaConf (and the reference's configs) mean a float."""
    if isinstance(x, dict):
        return {k: _numify(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_numify(v) for v in x]
    if isinstance(x, str) and _NUM.match(x):
        return float(x)
    return x


def _set_dotted(d: dict, key: str, value):
    parts = key.split(".")
    for p in parts[:-1]:
        d = d.setdefault(p, {})
    d[parts[-1]] = value


def load_config(argv, config_dir=None) -> Cfg:
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--config-name", default="desta25")
    ns, rest = ap.parse_known_args(argv)
    config_dir = config_dir or os.path.join(HERE, "config")
    path = os.path.join(config_dir, ns.config_name + ".yaml")
    if not os.path.isfile(path):
        raise FileNotFoundError(f"config '{ns.config_name}' not found under {config_dir} (pass --config-name)")
    with open(path) as f:
        cfg = yaml.safe_load(f) or {}
    for tok in rest:
        if "=" not in tok:
            raise ValueError(f"unrecognised argument '{tok}' (expected key=value or +group=name)")
        k, v = tok.split("=", 1)
        group = os.path.join(config_dir, k[1:], v + ".yaml") if k.startswith("+") and "." not in k else None
        if group is not None:                                  # +dataset=NAME -> merge config/dataset/NAME.yaml under 'dataset'
            with open(group) as f:
                cfg[k[1:]] = yaml.safe_load(f)
        elif k.startswith("+"):                                # Hydra's append syntax: +model.orca.enabled=true adds a key the YAML does not have
            _set_dotted(cfg, k[1:], yaml.safe_load(v))
        else:
            _set_dotted(cfg, k, yaml.safe_load(v))
    cfg = _numify(cfg)

    def check(d, prefix=""):
        for k, v in d.items():
            if isinstance(v, dict):
                check(v, prefix + k + ".")
            elif v == "???":
                raise ValueError(f"Missing mandatory value: {prefix}{k}")
    check(cfg)
    return Cfg(cfg)


def create_model(cfg: Cfg, device="cuda:0"):
    """Hydra config -> DeSTA25Config -> model (reference train_desta.py:96-130)."""
    from desta.models.modeling_desta25 import DeSTA25AudioModel, DeSTA25Config
    from desta.synthetic import RandomWeights
    orca = cfg.model.get("orca", {}) or {}
    llm, enc, con = cfg.model.llm, cfg.model.encoder, cfg.model.connector
    model_config = DeSTA25Config(
        llm_model_id=llm.model_id, encoder_model_id=enc.model_id, connector_mode=con.mode,
        qformer_num_hidden_layers=con.num_hidden_layers, prompt_size=con.prompt_size,
        use_lora=llm.get("use_lora", False), audio_locator=cfg.model.audio_locator,
        placeholder_token=cfg.model.placeholder_token,
        # model.orca.* -> orca_* with the reference's own fallbacks (reference train_desta.py:110-124; note kernel 7 / scale 5.0 here
        # against 5 / 2.5 in DeSTA25Config's signature)
        orca_enabled=orca.get("enabled", False), orca_local_enabled=orca.get("local_enabled", True),
        orca_global_cross_attn=orca.get("global_cross_attn", False), orca_deep_injection_enabled=orca.get("deep_injection_enabled", True),
        orca_audio_position_scale=orca.get("audio_position_scale", 5.0), orca_global_num_tokens=orca.get("global_num_tokens", 4),
        orca_local_downsample=orca.get("local_downsample", 4), orca_local_kernel_size=orca.get("local_kernel_size", 7),
        orca_gate_init=orca.get("gate_init", 0.1), orca_ortho_weight_global=orca.get("ortho_weight_global", 0.01),
        orca_ortho_diversity_weight=orca.get("ortho_diversity_weight", 0.01),
        orca_ortho_weight_qformer_local=orca.get("ortho_weight_qformer_local", 0.01), orca_align_weight_local=orca.get("align_weight_local", 0.05),
        llm_config=dict(llm.config) if llm.get("config") else None,
        encoder_config=dict(enc.config) if enc.get("config") else None,
        qformer_intermediate_size=con.get("intermediate_size", 3072))
    random_init = bool(llm.get("config")) or bool(llm.get("random_init", False))
    weights = RandomWeights(model_config, device, seed=0) if random_init else None     # no checkpoints offline
    model = DeSTA25AudioModel(model_config, weights=weights, device=device)
    model.config.train_id = 30678
    return model


def create_training_args(cfg: Cfg):
    """Reference train_desta.py:133-162, restricted to what the hot path consumes."""
    from desta.trainer.desta_trainer import TrainingArguments
    return TrainingArguments(
        output_dir=cfg.exp_dir, num_train_epochs=cfg.trainer.max_epochs,
        per_device_train_batch_size=cfg.dataset.train_ds.batch_size,
        gradient_accumulation_steps=cfg.trainer.accumulate_grad_batches,
        learning_rate=float(cfg.optim.lr), weight_decay=float(cfg.optim.weight_decay),
        warmup_steps=cfg.optim.sched.warmup_steps, logging_steps=cfg.trainer.log_every_n_steps,
        max_steps=cfg.trainer.get("max_steps", -1), bf16="bf16" in cfg.trainer.precision, optim="adafactor",
        save_strategy="epoch" if cfg.trainer.get("enable_checkpointing", False) else "no",
        eval_strategy="steps" if isinstance(cfg.trainer.get("val_check_interval"), int) else "epoch",      # reference :147-148
        eval_steps=cfg.trainer.get("val_check_interval") if isinstance(cfg.trainer.get("val_check_interval"), int) else None,
        per_device_eval_batch_size=cfg.dataset.validation_ds.batch_size,
        dataloader_num_workers=int(cfg.dataset.train_ds.get("num_workers", 0) or 0),        # reference :158
        dataloader_pin_memory=bool(cfg.dataset.train_ds.get("pin_memory", True)),           # reference :159
        overlap_comm=bool(cfg.trainer.get("overlap_comm", True)),
        overlap_encoder=bool(cfg.trainer.get("overlap_encoder", True)),
        overlap_connector_backward=bool(cfg.trainer.get("overlap_connector_backward", False)),
        # synthetic streams have no len(): every rank draws num_samples // batch_size batches per epoch
        steps_per_epoch=(cfg.dataset.train_ds.num_samples // cfg.dataset.train_ds.batch_size
                         if cfg.dataset.train_ds.get("synthetic", False) else None))


def load_pretrained_weights(model, path: str) -> None:
    """Lightning-style {'state_dict': ...} with a 'model.' prefix (reference :73-83); tensors only."""
    import torch
    blob = torch.load(path, map_location="cpu", weights_only=True)
    sd = blob.get("state_dict", blob)
    sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}
    model.load_state_dict(sd, strict=False)


def create_tokenizer(cfg: "Cfg"):
    """reference `_setup_generation` (modeling_desta25.py:1465-1470): AutoTokenizer of the LLM by name; offline that is the local
    HF cache (`HF_HOME`), with `HF_HUB_OFFLINE=1` semantics."""
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(cfg.model.llm.model_id, cache_dir=os.getenv("HF_HOME"), local_files_only=True)


def create_datasets(cfg: "Cfg", model, rank: int):
    """(train_dataset, eval_dataset, data_collator, tokenizer) as the reference builds them (train_desta.py:196-214)."""
    if cfg.dataset.train_ds.get("synthetic", False):
        return SyntheticAudioTextDataset(cfg, cfg.dataset.train_ds, model, rank), None, None, None
    from desta.trainer.data.simple_dataset import BaseAudioTextDataset
    model._setup_generation(tokenizer=create_tokenizer(cfg))
    tok, proc = model.tokenizer, model.processor
    train_ds = BaseAudioTextDataset(cfg, cfg.dataset.train_ds, tok, proc)
    val_cfg = cfg.dataset.get("validation_ds")
    eval_ds = BaseAudioTextDataset(cfg, val_cfg, tok, proc) if val_cfg and val_cfg.get("manifest_filepaths") else None
    return train_ds, eval_ds, train_ds.collate_fn, tok


class SyntheticAudioTextDataset:
    """Stands in for BaseAudioTextDataset: yields collated batches (simple_dataset.py:248-264 layout)."""

    def __init__(self, cfg: Cfg, data_cfg: Cfg, model, rank: int = 0):
        if not data_cfg.get("synthetic", False):
            raise ValueError("SyntheticAudioTextDataset needs `synthetic: true` (manifests go through create_datasets / BaseAudioTextDataset)")
        self.data_cfg, self.model, self.rank = data_cfg, model, rank
        S = data_cfg.context_tokens + model.config.prompt_size + data_cfg.target_tokens
        if S > data_cfg.max_seq_length:
            raise ValueError(f"sequence {S} exceeds max_seq_length {data_cfg.max_seq_length} (hazard H10)")

    def batches(self, epoch: int = 0):
        from desta import _hip
        from desta.synthetic import synthetic_inputs, synthetic_waveform
        dc, cfgm, dev = self.data_cfg, self.model.config, self.model.device
        for i in range(dc.num_samples // dc.batch_size):
            seed = 1234 + self.rank + 7919 * i + 104729 * epoch               # a pure function of (rank, epoch, index): resumable
            b = synthetic_inputs(cfgm, dc.batch_size, dc.context_tokens, dc.target_tokens, dev, seed=seed)
            wave = synthetic_waveform(dc.batch_size, dev, seed=seed)
            b["batch_features"] = _hip.logmel(wave, cfgm.encoder_config.num_mel_bins)
            yield b


def main(argv=None):
    import torch
    import torch.distributed as dist
    cfg = load_config(sys.argv[1:] if argv is None else argv)
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format="%(asctime)s %(levelname)s %(message)s")
    os.makedirs(cfg.exp_dir, exist_ok=True)
    if cfg.get("resume_from_checkpoint") and cfg.get("init_from_pretrained_weights"):
        raise AssertionError("Cannot provide both resume_from_checkpoint and init_from_pretrained_weights")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    from desta.trainer.desta_trainer import DeSTA25Trainer
    model = create_model(cfg, device=f"cuda:{local}")
    if cfg.get("init_from_pretrained_weights"):
        load_pretrained_weights(model, cfg.init_from_pretrained_weights)
    train_ds, eval_ds, collate, tok = create_datasets(cfg, model, rank)
    trainer = DeSTA25Trainer(model=model, args=create_training_args(cfg), cfg=cfg, train_dataset=train_ds, eval_dataset=eval_ds,
                             data_collator=collate, processing_class=tok)
    if rank == 0:
        with open(os.path.join(cfg.exp_dir, "config.yaml"), "w") as f:
            yaml.safe_dump(dict(cfg), f)
    if not cfg.get("resume_from_checkpoint"):
        # reference :220-228 — "eval before train to catch logic errors early", then checkpoint-initial
        if eval_ds is not None:
            logging.info("Running initial evaluation to verify model and trainer logic...")
            trainer.evaluate()
        if rank == 0:
            trainer.save_model(os.path.join(cfg.exp_dir, "checkpoint-initial"))
    # reference :231 — trainer.train(resume_from_checkpoint=...): parameters, optimizer.pt, scheduler.pt, step
    losses = trainer.train(resume_from_checkpoint=cfg.get("resume_from_checkpoint") or None)
    if losses:
        logging.info(f"{len(losses)} steps (global step {trainer.global_step} of {trainer.total_steps}), "
                     f"loss {losses[0]:.4f} -> {losses[-1]:.4f}")
    if world > 1:
        dist.destroy_process_group()
    return trainer


if __name__ == "__main__":
    main()
