"""Producer of the batch layout the hot path consumes.

Host-side mirror of the reference's `desta/trainer/data/simple_dataset.py`:

* `prepare_audio_context_and_start_positions` / `prepare_audio_context_with_start_end_tags` — placeholder expansion of the
  `<|AUDIO|>` locator (`modeling_desta25.py:99-123`, `simple_dataset.py:42-101`);
* `BaseCollateFn` — left-padded tokenisation, labels, `start_answer_position`, pad-shifted `batch_start_positions`, the
  `[1, n]` transcription ids, the context-only copy for evaluation and the `_empty_batch` marker (`:116-301`);
* `BaseAudioTextDataset` — prompt-only preprocessing of JSONL manifests (`:574-743`) behind the reference's disk cache
  (`:361-452`): `$HF_HOME/desta_preprocessed/<md5 of the manifest paths>` written with `datasets.save_to_disk` by rank 0
  under a `.lock` file and published by a `.ready` file, the other ranks wait at the barrier and load it; `records=` (tests)
  or `disk_cache=False` preprocess in memory.

All of it is integer / index / string work on the host; it takes ANY object with the tokenizer call protocol (`__call__`
with padding / truncation, `tokenize`, `encode`, `convert_tokens_to_string`, `apply_chat_template`, `padding_side`,
`eos_token`) and any `processor(list_of_waveforms, sampling_rate=16000, return_tensors="pt").input_features`.  The MI355X
difference is the processor: `desta.utils.audio.HipLogMelProcessor` runs the log-mel kernel on the device instead of the CPU
`WhisperFeatureExtractor`, and waveforms stay numpy / torch arrays (the reference round-trips them through `tolist()`).
Bit parity of every integer field against the reference's own class: tests/test_collate.py (fixture made by importing the
reference, tests/golden/make_collate_golden.py).
"""
from __future__ import annotations

import hashlib
import json
import logging
import os
import re
import time
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from ...utils.audio import AudioSegment

START_TAG, END_TAG = "<start_audio>", "<end_audio>"


def _get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _is_main_process() -> bool:
    return _get_rank() == 0


def _barrier() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def resolve_manifest_filepath(fp: str) -> str:
    """The reference resolves manifests with `lulutils.resolve_filepath` (third-party, absent offline: parity of the resolved
    STRING, and with it of the cache key, is unpinned).  Here: `~` / environment variables expanded, absolute, must exist."""
    out = os.path.abspath(os.path.expandvars(os.path.expanduser(fp)))
    if not os.path.isfile(out):
        raise FileNotFoundError(f"manifest not found: {fp}")
    return out


def prepare_audio_context_and_start_positions(token_list: Sequence[str], audio_locator: str, audio_size_list: List[int],
                                              transcription_size_list: List[int], placeholder_token: str) -> Tuple[List[str], List[int]]:
    """Every `audio_locator` token becomes audio_size + transcription_size placeholder tokens; the index of the first one
    is the audio's start position (`modeling_desta25.py:99-123`).  The size lists are consumed front to back."""
    if len(audio_size_list) != len(transcription_size_list):
        raise AssertionError(f"audio_size_list and transcription_size_list must have the same length, audio_size_list: "
                             f"{audio_size_list}, transcription_size_list: {transcription_size_list}")
    out: List[str] = []
    starts: List[int] = []
    k = 0                                                             # locators seen == entries consumed from the FRONT of both lists
    for tok in token_list:
        if tok != audio_locator:
            out.append(tok)
            continue
        if k >= len(audio_size_list):
            raise IndexError("pop from empty list")                   # more locators than audios (what list.pop(0) raises)
        starts.append(len(out))
        out += [placeholder_token] * (audio_size_list[k] + transcription_size_list[k])
        k += 1
    del audio_size_list[:k], transcription_size_list[:k]              # the reference pops them; callers see the lists shrink
    return out, starts


def prepare_audio_context_with_start_end_tags(text: str, audio_size_list: List[int], transcription_size_list: List[int],
                                              placeholder_token: str, tokenizer, start_tag: str = START_TAG,
                                              end_tag: str = END_TAG) -> Tuple[str, List[int]]:
    """`<start_audio>…<end_audio>` blocks -> placeholder runs (`simple_dataset.py:42-101`)."""
    blocks = list(re.finditer(re.escape(start_tag) + r".*?" + re.escape(end_tag), text, re.DOTALL))
    if len(blocks) != len(audio_size_list):
        logging.warning(f"Audio block count ({len(blocks)}) != audio_size_list ({len(audio_size_list)})")
    toks: List[str] = []
    starts: List[int] = []
    pos = 0
    n_sized = min(len(audio_size_list), len(transcription_size_list))
    for j, m in enumerate(blocks):
        if m.start() > pos:
            toks += tokenizer.tokenize(text[pos:m.start()], add_special_tokens=False)
        starts.append(len(toks))
        if j < n_sized:
            toks += [placeholder_token] * (audio_size_list[j] + transcription_size_list[j])
        pos = m.end()
    if pos < len(text):
        toks += tokenizer.tokenize(text[pos:], add_special_tokens=False)
    return tokenizer.convert_tokens_to_string(toks), starts


def resolve_audio_filepath(path: str) -> str:
    """The path itself, or the same stem with `.wav` (`simple_dataset.py:104-114`)."""
    if os.path.exists(path):
        return path
    wav = os.path.splitext(path)[0] + ".wav"
    if os.path.exists(wav):
        return wav
    raise FileNotFoundError(f"Audio file not found: {path}")


class BaseCollateFn:
    """List of preprocessed samples -> the batch dict `DeSTA25AudioModel.forward` / `_generate_step` consume."""

    def __init__(self, data_cfg, tokenizer, processor, audio_loader=None):
        self.tokenizer, self.processor = tokenizer, processor
        self.max_seq_length = data_cfg["max_seq_length"] if isinstance(data_cfg, dict) else data_cfg.max_seq_length
        # `audio` entries are file paths (decoded by desta.utils.audio) or already-decoded waveforms
        self.audio_loader = audio_loader or (lambda a: AudioSegment.from_file(a, target_sr=16000, channel_selector="average").samples)

    def _tok(self, texts: List[str]):
        return self.tokenizer(texts, truncation=True, padding="longest", max_length=self.max_seq_length, return_tensors="pt",
                              return_length=True, add_special_tokens=False)

    def __call__(self, batch: List[Dict[str, Any]]) -> Dict[str, Any]:
        """The reference's one-call collate (`simple_dataset.py:130-301`) = the host half + the device half below."""
        return self.finish(self.host_collate(batch))

    def host_collate(self, batch: List[Dict[str, Any]]) -> Dict[str, Any]:
        """Everything of the collate that needs no device: audio decode + resample, the two tokenisations, the index arithmetic
        and the decoded clips as a list of 1-D float32 tensors `_waves`.  This is what runs in the DataLoader WORKER processes (`dataset.train_ds.num_workers`,
        reference: examples/train/train_desta.py:158-159 -> HF `dataloader_num_workers`); the reference's workers also run the CPU
        log-mel there, here the log-mel is a device kernel and is applied by `finish` in the training process."""
        tok = self.tokenizer
        assert tok.padding_side == "left", f"padding_side must be left, got {tok.padding_side}"
        # 1. decode; a sample with any undecodable audio is dropped, an all-bad batch becomes the empty marker
        kept, waves = [], []
        for item in batch:
            ws = []
            for a in item["processed_audios"]:
                try:
                    ws.append(self.audio_loader(a["audio"]))
                except Exception as e:                                 # noqa: BLE001 (the reference catches everything here)
                    logging.warning(f"Skipping sample due to audio decode error: {a['audio']} - {e}")
                    ws = None
                    break
            if ws is not None:
                kept.append(item)
                waves.append(ws)
        if not kept:
            failed = [a.get("audio", "unknown") for item in batch for a in item.get("processed_audios", [])]
            logging.warning(f"Entire batch skipped due to audio decode errors. Failed paths: {failed[:3]}...")
            return {"_empty_batch": True}
        batch = kept
        # 2. two left-padded tokenisations: context + target (training) and context only (generation)
        full = self._tok([it["audio_context"] + it["target"] for it in batch])
        ctx = self._tok([it["audio_context"] for it in batch])
        ids, mask = full["input_ids"], full["attention_mask"]
        # 3. index arithmetic, vectorised: pad = padded length - real tokens; the answer starts after pad + context tokens
        n_ctx_tokens = torch.tensor([len(tok.tokenize(it["audio_context"])) for it in batch], dtype=torch.long)
        pad = torch.as_tensor(full["length"], dtype=torch.long) - mask.sum(dim=1)
        ctx_pad = torch.as_tensor(ctx["length"], dtype=torch.long) - ctx["attention_mask"].sum(dim=1)
        answer_start = pad + n_ctx_tokens
        col = torch.arange(ids.shape[1]).unsqueeze(0)
        labels = torch.where(col >= answer_start.unsqueeze(1), ids, torch.full_like(ids, -100))
        features, starts, ctx_starts, tr_ids = [], [], [], []
        for i, it in enumerate(batch):
            features += waves[i]
            tr_ids += [tok.encode(t, add_special_tokens=False, return_tensors="pt").long() for t in it["transcription_list"]]
            starts += [(i, s + pad[i]) for s in it["start_positions"]]
            ctx_starts += [(i, s + ctx_pad[i]) for s in it["start_positions"]]
        import numpy as np
        assert len(features) == len(starts) == len(tr_ids), \
            f"Length mismatch: features={len(features)}, positions={len(starts)}, transcriptions={len(tr_ids)}"
        out = {"input_ids": ids, "attention_mask": mask, "labels": labels,
               "audio_start_answer_positions": list(answer_start.unbind(0)),
               # the decoded waveforms as 1-D float32 tensors (zero-copy views: tensors travel from worker processes through shared
               # memory and can be pinned by the loader); the processor still receives them as the reference's list of clips
               "_waves": [w if isinstance(w, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)) for w in features],
               "batch_transcription_ids": tr_ids, "batch_start_positions": starts,
               "context_input_ids": ctx["input_ids"], "context_attention_mask": ctx["attention_mask"],
               "context_batch_start_positions": ctx_starts, "metadata": list(batch)}
        # optional ORCA prosody side inputs are carried through untouched in shape (zero-filled where a sample has none)
        if any("f0_energy_global" in it for it in batch):
            out["f0_energy_global"] = torch.stack([torch.tensor(it["f0_energy_global"], dtype=torch.float32) if "f0_energy_global" in it
                                                   else torch.zeros(4) for it in batch])
        if any("f0_energy_local" in it for it in batch):
            loc = [torch.tensor(it["f0_energy_local"], dtype=torch.float32) if "f0_energy_local" in it else None for it in batch]
            T = max(t.shape[0] for t in loc if t is not None)
            out["f0_energy_local"] = torch.stack([torch.zeros(T, 2) if t is None else torch.nn.functional.pad(t, (0, 0, 0, T - t.shape[0]))
                                                  for t in loc])
        return out

    def finish(self, partial: Dict[str, Any]) -> Dict[str, Any]:
        """Device half: `processor(...)` on the padded clips (`simple_dataset.py:239-243`) -> `batch_features`."""
        if partial.get("_empty_batch", False) or "_waves" not in partial:
            return partial
        out = dict(partial)
        waves = out.pop("_waves")
        feats = self.processor(waves, sampling_rate=16000, return_tensors="pt").input_features
        assert len(feats) == len(out["batch_start_positions"]) == len(out["batch_transcription_ids"]), \
            f"Length mismatch: features={len(feats)}, positions={len(out['batch_start_positions'])}, transcriptions={len(out['batch_transcription_ids'])}"
        # key order of the reference's dict (`:248-264`): batch_features sits between the answer positions and the transcription ids
        order = ["input_ids", "attention_mask", "labels", "audio_start_answer_positions", "batch_features"]
        out["batch_features"] = feats
        return {**{k: out[k] for k in order}, **{k: v for k, v in out.items() if k not in order}}


class BaseAudioTextDataset:
    """JSONL manifests -> preprocessed samples (`audio_context`, `start_positions`, `processed_audios`, `transcription_list`,
    `target`, `length`).  Training is prompt-only (hazard H9): `messages` / `seed_description` are ignored, the audio locator is
    appended to the prompt when missing, transcriptions are always "" so an audio span is exactly `prompt_size` tokens."""

    READY_POLL_S, READY_TIMEOUT_S = 5, 7200                                         # `simple_dataset.py:438-452`

    def __init__(self, cfg, data_cfg, tokenizer, processor, records: Optional[List[Dict[str, Any]]] = None,
                 disk_cache: bool = True):
        g = (lambda o, k, d=None: (o.get(k, d) if hasattr(o, "get") else getattr(o, k, d)))
        model_cfg = g(cfg, "model")
        self.audio_locator, self.placeholder_token = g(model_cfg, "audio_locator"), g(model_cfg, "placeholder_token")
        con = g(model_cfg, "connector")
        self.prompt_size, self.connector_mode = g(con, "prompt_size"), g(con, "mode")
        self.orca_global_num_tokens = g(g(model_cfg, "orca", {}) or {}, "global_num_tokens", 4)
        self.system_prompt = g(model_cfg, "system_prompt", None)
        self.data_root = g(data_cfg, "data_root", "")
        self.tokenizer, self.processor = tokenizer, processor
        if records is None:
            paths = g(data_cfg, "manifest_filepaths")
            self.manifest_filepaths = [paths] if isinstance(paths, str) else list(paths)
            for fp in self.manifest_filepaths:
                logging.info(f"Loading manifest: {fp}")
            data_files = [resolve_manifest_filepath(fp) for fp in self.manifest_filepaths]
            if disk_cache:
                rows = self._load_or_preprocess_on_disk(data_files)
            else:
                records = []
                for fp in data_files:
                    with open(fp) as f:
                        records += [json.loads(line) for line in f if line.strip()]
        if records is not None:
            rows = self._preprocess_records(records)
        n = len(rows)
        valid = (lambda r: r["length"] > 0 and len(r["audio_context"]) > 0 and len(r["processed_audios"]) > 0)
        self.dataset = rows.filter(valid) if hasattr(rows, "filter") else [r for r in rows if valid(r)]
        logging.info(f"Dataset: {n} samples, {len(self.dataset)} valid, {n - len(self.dataset)} skipped")
        self.collate_fn = BaseCollateFn(data_cfg=data_cfg, tokenizer=tokenizer, processor=processor)

    def _preprocess_records(self, records: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
        cols = {k: [r.get(k) for r in records] for k in ("id", "prompt", "response")}
        cols["prompt"] = [p or "" for p in cols["prompt"]]
        cols["response"] = [r or "" for r in cols["response"]]
        pre = self._preprocess_function(cols)
        return [{k: pre[k][i] for k in pre} for i in range(len(records))]

    @staticmethod
    def cache_paths(data_files: List[str]) -> Tuple[str, str, str]:
        """(cache_dir, lock_file, ready_file) of a manifest set — the reference's layout (`simple_dataset.py:367-376`)."""
        key = hashlib.md5("_".join(sorted(data_files)).encode()).hexdigest()[:12]
        cache_dir = os.path.join(os.environ.get("HF_HOME", os.path.expanduser("~/.cache/huggingface")), "desta_preprocessed", key)
        return cache_dir, cache_dir + ".lock", cache_dir + ".ready"

    def _load_or_preprocess_on_disk(self, data_files: List[str]):
        """`simple_dataset.py:361-452`: a ready cache is loaded by every rank; otherwise rank 0 preprocesses under a lock file,
        saves with `datasets.save_to_disk`, publishes the `.ready` file and removes the lock; everybody meets at the barrier and
        the other ranks poll for `.ready` (5 s steps, 2 h) and load.  A cache that fails to load is re-made."""
        import datasets
        cache_dir, lock_file, ready_file = self.cache_paths(data_files)
        ds = None
        if os.path.exists(ready_file) and os.path.isdir(cache_dir):
            logging.info(f"[Rank {_get_rank()}] Loading preprocessed dataset from disk cache: {cache_dir}")
            try:
                ds = datasets.load_from_disk(cache_dir)
            except Exception as e:                                                    # noqa: BLE001 (the reference re-makes on ANY failure)
                logging.warning(f"[Rank {_get_rank()}] Cache load failed: {e}. Will reprocess.")
                if os.path.exists(ready_file):
                    os.remove(ready_file)
        if ds is not None:
            return ds
        if _is_main_process():
            os.makedirs(os.path.dirname(lock_file), exist_ok=True)
            with open(lock_file, "w") as f:
                f.write(f"rank0_processing_{os.getpid()}")
            try:
                # the JSON builder's own arrow cache next to ours, under the HF_HOME of THIS call (`datasets` froze its default
                # at import time)
                raw = datasets.load_dataset("json", data_files=data_files,
                                            cache_dir=os.path.join(os.path.dirname(os.path.dirname(cache_dir)), "datasets"))["train"]
                ds = raw.map(self._preprocess_function, batched=True, batch_size=128, num_proc=1, load_from_cache_file=False,
                             keep_in_memory=False)
                ds.save_to_disk(cache_dir)
                with open(ready_file, "w") as f:
                    f.write("ready")
                logging.info(f"[Rank {_get_rank()}] Preprocessing complete. Saved {len(ds)} samples.")
            finally:
                if os.path.exists(lock_file):
                    os.remove(lock_file)
        _barrier()
        if not _is_main_process():
            waited = 0
            while not os.path.exists(ready_file) and waited < self.READY_TIMEOUT_S:
                time.sleep(self.READY_POLL_S)
                waited += self.READY_POLL_S
            if not os.path.exists(ready_file):
                raise RuntimeError(f"[Rank {_get_rank()}] Timeout waiting for preprocessed dataset!")
            ds = datasets.load_from_disk(cache_dir)
        return ds

    def _preprocess_function(self, examples: Dict[str, List]) -> Dict[str, List]:
        tok = self.tokenizer
        ids = examples["id"]
        prompts = examples.get("prompt", [""] * len(ids))
        n = len(ids)
        ctxs, starts_l, audios_l, trans_l = [""] * n, [[] for _ in range(n)], [[] for _ in range(n)], [[] for _ in range(n)]
        block_re = re.compile(re.escape(START_TAG) + r".*?" + re.escape(END_TAG), re.DOTALL)
        for i, (sid, prompt) in enumerate(zip(ids, prompts)):
            text = (prompt or "").strip()
            if not text:
                continue                                                               # empty prompt: skipped
            content = text if self.audio_locator in text else f"{text} {self.audio_locator}"
            messages = ([{"role": "system", "content": self.system_prompt}] if self.system_prompt else [])
            messages.append({"role": "user", "content": content, "audios": [{"audio": sid, "text": ""}]})
            ctx = tok.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
            try:
                audios = [{"audio": resolve_audio_filepath(os.path.join(self.data_root, sid)), "text": ""}]
            except FileNotFoundError:
                continue                                                               # audio file missing: skipped
            size = self.orca_global_num_tokens if self.connector_mode == "orca_hybrid" else self.prompt_size
            sizes, trans = [size] * len(audios), [""] * len(audios)
            tsizes = [len(tok.tokenize(t, add_special_tokens=False)) for t in trans]
            if block_re.search(ctx):
                ctx, st = prepare_audio_context_with_start_end_tags(ctx, sizes, tsizes, self.placeholder_token, tok)
            elif ctx.count(self.audio_locator) > 0:
                toks, st = prepare_audio_context_and_start_positions(tok.tokenize(ctx), self.audio_locator, sizes, tsizes, self.placeholder_token)
                ctx = tok.convert_tokens_to_string(toks)
            else:
                continue                                                               # no audio marker survived the template
            ctxs[i], starts_l[i], audios_l[i], trans_l[i] = ctx, st, audios, trans
        examples["audio_context"], examples["start_positions"] = ctxs, starts_l
        examples["transcription_list"], examples["processed_audios"] = trans_l, audios_l
        targets, lengths = [], []
        for ctx, resp in zip(ctxs, examples.get("response", [""] * n)):
            ok = bool(ctx) and bool(resp)
            targets.append(resp + tok.eos_token if ok else "")
            lengths.append(len(tok.tokenize(ctx + resp)) if ok else 0)
        examples["target"], examples["length"] = targets, lengths
        return examples

    def __len__(self) -> int:
        return len(self.dataset)

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        return self.dataset[idx]
