"""GPU: the real data path end to end (SURVEY §8a A2 + §8f-3): WAVE files -> `BaseAudioTextDataset` preprocessing ->
`BaseCollateFn` with the DEVICE log-mel processor -> `model(**batch)`; loss / logits against the oracle fed with the same
collated integers and the oracle's own log-mel of the same decoded clips."""
import io
import os
import wave

import numpy as np
import pytest
import torch

import desta_oracle as O
from helpers import ToyTokenizer, cfg_from_dims, rel_err

pytestmark = pytest.mark.gpu


def _write_wav(path, x, sr):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1 if x.ndim == 1 else x.shape[1])
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


def test_wav_files_through_collate_into_the_model(tmp_path):
    from desta.models.modeling_desta25 import DeSTA25AudioModel
    from desta.trainer.data.simple_dataset import BaseAudioTextDataset
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments
    from desta.utils.audio import AudioSegment, HipLogMelProcessor, pad_or_trim
    rng = np.random.default_rng(0)
    t = np.arange(44100 * 3) / 44100
    _write_wav(tmp_path / "a.wav", np.stack([0.3 * np.sin(2 * np.pi * 330 * t), 0.2 * np.sin(2 * np.pi * 880 * t)], 1), 44100)   # stereo, 44.1 kHz
    _write_wav(tmp_path / "b.wav", 0.1 * rng.standard_normal(16000 * 2), 16000)                                                     # mono, 16 kHz
    _write_wav(tmp_path / "c.wav", 0.1 * rng.standard_normal(8000 * 35), 8000)                                                     # 35 s: trimmed to 30 s
    (tmp_path / "broken.wav").write_bytes(b"RIFFxxxx")
    records = [dict(id="a.wav", prompt="Describe the audio.", response="two tones , one low one high ."),
               dict(id="b", prompt="What do you hear? <|AUDIO|> Be brief.", response="noise"),      # id without extension -> .wav twin
               dict(id="c.wav", prompt="And this one", response="more noise for a long time"),
               dict(id="broken.wav", prompt="undecodable", response="dropped by the collate function")]
    d = O.tiny_dims(False)
    d.enc_T = 1500                                                     # real 30 s clips: 3000 mel frames
    cfg = {"model": {"audio_locator": "<|AUDIO|>", "placeholder_token": "<|video_pad|>", "connector": {"prompt_size": 64, "mode": "qformer_1"}}}
    tok = ToyTokenizer(vocab_size=d.vocab)
    ds = BaseAudioTextDataset(cfg, {"data_root": str(tmp_path), "max_seq_length": 512}, tok, HipLogMelProcessor(d.n_mels), records=records)
    assert len(ds) == 4
    batch = ds.collate_fn([ds[i] for i in range(4)])                   # broken.wav is dropped inside the collate function
    assert batch["input_ids"].shape[0] == 3 and batch["batch_features"].shape == (3, d.n_mels, 3000) and batch["batch_features"].is_cuda
    assert ds.collate_fn([ds[3]]) == {"_empty_batch": True}
    # oracle: same integers, its own log-mel of the same decoded / resampled / padded clips
    waves = [AudioSegment.from_file(m["processed_audios"][0]["audio"], target_sr=16000, channel_selector="average").samples for m in batch["metadata"]]
    assert waves[0].shape[0] == 48000 and waves[2].shape[0] == 560000
    mel_o = O.logmel(pad_or_trim(waves), d.n_mels)
    assert float((batch["batch_features"].cpu() - mel_o).abs().max()) < 5e-4
    w = O.init_weights(d, seed=7)
    model = DeSTA25AudioModel(cfg_from_dims(d), weights=w)
    ob = {"input_ids": batch["input_ids"], "attention_mask": batch["attention_mask"], "labels": batch["labels"], "batch_features": mel_o,
          "batch_start_positions": [(int(i), int(s)) for i, s in batch["batch_start_positions"]],
          "batch_transcription_ids": batch["batch_transcription_ids"]}
    loss_o, logits_o = O.model_forward(w, d, ob)
    model.eval()
    out = model(**batch)
    m = batch["attention_mask"].bool()
    assert abs(float(out.loss) - float(loss_o)) < 2e-2 and rel_err(out.logits.float().cpu()[m], logits_o[m]) < 3e-2
    # and one optimizer step through the trainer with the collated batch (+ generation from its context half)
    tr = DeSTA25Trainer(model, args=TrainingArguments(learning_rate=1e-3, warmup_steps=0, max_steps=4, logging_steps=1), processing_class=tok)
    before = model.arena.params.clone()
    l0 = float(tr.training_step(batch))
    tr.wait_update()
    assert abs(l0 - float(loss_o)) < 3e-2 and not torch.equal(before, model.arena.params)
    ids = tr._predict_step(batch, {"max_new_tokens": 4})
    assert ids.shape == (3, 4) and len(tr.prediction_step_outputs) == 3 and "prediction" in tr.prediction_step_outputs[0]


def test_trainer_with_dataloader_workers_equals_inline(tmp_path):
    """`dataset.train_ds.num_workers` on the device path: 4 optimizer steps of `DeSTA25Trainer.train()` over WAVE files with the
    collate's host half in 2 forked worker processes (+ pinned hand-over, device log-mel in the training process, encoder prefetch
    on its own stream) end with bit-identical losses and parameters to the same run with the collate inline."""
    from desta.models.modeling_desta25 import DeSTA25AudioModel
    from desta.synthetic import WordTokenizer, write_synthetic_wav_dataset
    from desta.trainer.data.simple_dataset import BaseAudioTextDataset
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments
    from desta.utils.audio import HipLogMelProcessor
    d = O.tiny_dims(False)
    d.enc_T = 1500
    recs = write_synthetic_wav_dataset(str(tmp_path), 8, seed=5)
    cfg = {"model": {"audio_locator": "<|AUDIO|>", "placeholder_token": "<|video_pad|>", "connector": {"prompt_size": 64, "mode": "qformer_1"}}}
    w = O.init_weights(d, seed=7)
    res = {}
    for nw in (0, 2):
        tok = WordTokenizer(vocab_size=d.vocab)
        ds = BaseAudioTextDataset(cfg, {"data_root": str(tmp_path), "max_seq_length": 512}, tok, HipLogMelProcessor(d.n_mels), records=recs)
        model = DeSTA25AudioModel(cfg_from_dims(d), weights=w)
        tr = DeSTA25Trainer(model, args=TrainingArguments(learning_rate=1e-3, warmup_steps=0, max_steps=4, logging_steps=1, per_device_train_batch_size=2,
                                                          dataloader_num_workers=nw, num_train_epochs=1),
                            train_dataset=ds, data_collator=ds.collate_fn, processing_class=tok)
        losses = tr.train()
        torch.cuda.synchronize()
        assert tr.global_step == 4
        res[nw] = (losses, model.arena.params.clone())
    assert res[0][0] == res[2][0], (res[0][0], res[2][0])
    assert torch.equal(res[0][1], res[2][1])
