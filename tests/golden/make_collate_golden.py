"""Golden vectors for SURVEY §8a row A2 (batch layout producer): run the REFERENCE's own `BaseAudioTextDataset.
_preprocess_function` and `BaseCollateFn.__call__` (desta/trainer/data/simple_dataset.py:130-301, 574-743) on toy records with
the duck-typed `ToyTokenizer` of tests/helpers.py and store every integer / string field they produce as JSON.

Runs only in the build container.  What was needed to import the reference module here (ordinary ModuleNotFoundError, nothing
was denied): `omegaconf`, `lulutils`, `librosa`, `soundfile`, `pydub` are absent -> empty stub modules (`DictConfig = dict`,
`resolve_filepath = identity`); audio FILE decode is out of scope, so `simple_dataset.AudioSegment` is replaced by a fake whose
`from_file(path, ...)` returns a seeded waveform for known keys and raises for the keys listed as undecodable; the dataset
object is made with `__new__` (the constructor reads manifests through HF `datasets` + a lock-file protocol).  The processor is
a stub that records what it was called with (log-mel parity is pinned separately, row A1).
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))


def _import_reference():
    # before stubbing: transformers probes optional packages with find_spec when these lazy names are resolved
    from transformers import AutoFeatureExtractor, AutoTokenizer, AutoProcessor, WhisperForConditionalGeneration  # noqa: F401
    from transformers import AutoModelForCausalLM, BertConfig, PreTrainedModel  # noqa: F401
    from transformers.models.bert.modeling_bert import BertEncoder  # noqa: F401
    import datasets  # noqa: F401
    for name in ("librosa", "soundfile", "pydub", "pydub.exceptions", "omegaconf", "lulutils"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["soundfile"].available_formats = lambda: {}
    sys.modules["pydub"].AudioSegment = object
    sys.modules["pydub.exceptions"].CouldntDecodeError = Exception
    sys.modules["librosa"].__path__ = []
    sys.modules["omegaconf"].DictConfig = dict
    sys.modules["lulutils"].resolve_filepath = lambda p: p
    sys.path.insert(0, REF)
    from desta.trainer.data import simple_dataset as SD
    return SD


class StubProcessor:
    def __call__(self, feats, sampling_rate=None, return_tensors=None):
        self.calls = getattr(self, "calls", []) + [dict(n=len(feats), lens=[len(f) for f in feats], sampling_rate=sampling_rate)]
        return types.SimpleNamespace(input_features=torch.zeros(len(feats), 2, 4))


def wave_for(key: str) -> np.ndarray:
    import zlib
    n = 1000 + zlib.crc32(key.encode()) % 500
    return np.random.default_rng(zlib.crc32(key.encode())).standard_normal(n).astype(np.float32)


def _jsonable(x):
    if isinstance(x, torch.Tensor):
        return x.tolist()
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if isinstance(x, dict):
        return {k: _jsonable(v) for k, v in x.items()}
    if isinstance(x, (np.integer,)):
        return int(x)
    return x


def main():
    from helpers import COLLATE_CASES, ToyTokenizer
    SD = _import_reference()
    out = {}
    for name, (records, batches, bad) in COLLATE_CASES.items():
        for system_prompt in (None, "Focus on the audio clips and instructions."):
            tok = ToyTokenizer()
            with tempfile.TemporaryDirectory() as root:
                for r in records:
                    if not r["id"].startswith("missing"):
                        open(os.path.join(root, r["id"]), "wb").close()
                ds = SD.BaseAudioTextDataset.__new__(SD.BaseAudioTextDataset)
                ds.audio_locator, ds.placeholder_token, ds.data_root = "<|AUDIO|>", "<|video_pad|>", root
                ds.prompt_size, ds.tokenizer, ds.connector_mode, ds.orca_global_num_tokens = 64, tok, "qformer_1", 4
                ds.system_prompt = system_prompt
                cols = {k: [r[k] for r in records] for k in ("id", "prompt", "response")}
                pre = ds._preprocess_function({k: list(v) for k, v in cols.items()})
                rows = [{k: pre[k][i] for k in pre} for i in range(len(records))]

                class FakeSeg:
                    @staticmethod
                    def from_file(path, target_sr=None, channel_selector=None, **kw):
                        key = os.path.basename(path)
                        if key in bad:
                            raise RuntimeError(f"cannot decode {key}")
                        return types.SimpleNamespace(samples=wave_for(key))
                SD.AudioSegment = FakeSeg
                case = {"preprocess": [{k: (v if k != "processed_audios" else [os.path.basename(a["audio"]) for a in v])
                                        for k, v in row.items() if k in ("audio_context", "start_positions", "transcription_list",
                                                                          "processed_audios", "target", "length")} for row in rows],
                        "collate": []}
                valid = [r for r in rows if r["length"] > 0 and len(r["audio_context"]) > 0 and len(r["processed_audios"]) > 0]
                for max_len in (4096, 90):
                    coll = SD.BaseCollateFn({"max_seq_length": max_len} if False else types.SimpleNamespace(max_seq_length=max_len), tok, StubProcessor())
                    for idx in batches:
                        items = [valid[i] for i in idx if i < len(valid)]
                        b = coll(items)
                        rec = {"max_seq_length": max_len, "items": [i for i in idx if i < len(valid)]}
                        if b.get("_empty_batch"):
                            rec["_empty_batch"] = True
                        else:
                            rec.update({k: _jsonable(b[k]) for k in ("input_ids", "attention_mask", "labels", "context_input_ids",
                                                                     "context_attention_mask")})
                            rec["audio_start_answer_positions"] = [int(x) for x in b["audio_start_answer_positions"]]
                            rec["batch_start_positions"] = [[int(i), int(s)] for i, s in b["batch_start_positions"]]
                            rec["context_batch_start_positions"] = [[int(i), int(s)] for i, s in b["context_batch_start_positions"]]
                            rec["batch_transcription_ids"] = [list(t.shape) + [str(t.dtype)] for t in b["batch_transcription_ids"]]
                            rec["n_features"] = int(b["batch_features"].shape[0])
                            rec["processor_calls"] = coll.processor.calls[-1]
                            rec["metadata_ids"] = [os.path.basename(m["processed_audios"][0]["audio"]) for m in b["metadata"]]
                        case["collate"].append(rec)
                out[f"{name}|system={'yes' if system_prompt else 'no'}"] = case
    # chat-level generate() (modeling_desta25.py:1491-1721): the reference's own method on an object assembled with __new__, the
    # front-end models stubbed (toy tokenizer, recording processor, a VAD that finds speech only in g1 / g2, an ASR stand-in
    # for the Whisper decoder), `_generate_step` replaced by a recorder -> the `inputs` dict it is handed is the golden
    from helpers import GENERATE_MESSAGES
    from desta.models import modeling_desta25 as M

    def wave(path, **kw):
        return types.SimpleNamespace(samples=wave_for(os.path.basename(path)))
    M.AudioSegment = types.SimpleNamespace(from_file=wave)
    with tempfile.TemporaryDirectory() as root:
        msgs = json.loads(json.dumps(GENERATE_MESSAGES))
        for conv in msgs:
            for m in conv:
                for a in m.get("audios", []):
                    a["audio"] = os.path.join(root, a["audio"])
                    open(a["audio"], "wb").close()
        captured = {}

        class R(M.DeSTA25AudioModel):
            device = torch.device("cpu")

            def _generate_step(self, inputs, **kw):
                captured["inputs"], captured["kw"] = inputs, kw
                return torch.tensor([[11, 12, 2], [13, 2, 0]])
        ref = R.__new__(R)
        torch.nn.Module.__init__(ref)
        tok = ToyTokenizer()
        tok.pad_token, tok.pad_token_id = tok.eos_token, tok.eos_token_id
        ref.tokenizer, ref.processor = tok, StubProcessor()
        ref.config = types.SimpleNamespace(prompt_size=64, connector_mode="qformer_1")
        ref.audio_locator, ref.placeholder_token = "<|AUDIO|>", "<|video_pad|>"
        ref.vad_model = object()
        ref.get_speech_timestamps = lambda feat, model: [1] if len(feat) != len(wave_for("g3.wav")) else []
        ref.perception = types.SimpleNamespace(whisper=types.SimpleNamespace(generate=lambda **kw: torch.tensor([[7, 8, 9]])))
        ref.processor.batch_decode = lambda ids, skip_special_tokens=True: [" spoken words "]
        outg = M.DeSTA25AudioModel.generate(ref, msgs, do_sample=False, max_new_tokens=3)
        gi = captured["inputs"]
        out["generate_with_audio"] = {
            "context_input_ids": gi["context_input_ids"].tolist(), "context_attention_mask": gi["context_attention_mask"].tolist(),
            "context_batch_start_positions": [[int(i), int(sp)] for i, sp in gi["context_batch_start_positions"]],
            "batch_transcription_ids": [t.tolist() for t in gi["batch_transcription_ids"]],
            "n_features": int(gi["batch_features"].shape[0]), "processor_calls": ref.processor.calls,
            "kw": {k: v for k, v in captured["kw"].items()}, "text": outg.text, "generated_ids": outg.generated_ids,
            "audios": [[os.path.basename(a), t] for a, t in outg.audios]}
    # placeholder expansion helper on its own (modeling_desta25.py:99-123), incl. two audios + transcription sizes
    from desta.models.modeling_desta25 import _prepare_audio_context_and_start_positions as prep
    toks = "a <|AUDIO|> b c <|AUDIO|> d".split()
    r, st = prep(list(toks), "<|AUDIO|>", [3, 2], [1, 0], "P")
    out["prepare_two_audios"] = {"tokens": toks, "sizes": [[3, 2], [1, 0]], "result": r, "starts": st}
    path = os.path.join(HERE, "ref_collate.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", {k: (len(v.get("collate", [])) if isinstance(v, dict) else 0) for k, v in out.items()})


if __name__ == "__main__":
    main()
