"""Decode-free part of the reference's audio front end (`desta/utils/audio.py:117-361`) + the device log-mel
processor that replaces the CPU `WhisperFeatureExtractor` in the collate function.

`AudioSegment.from_file(path, target_sr=16000, channel_selector="average").samples` keeps the reference's call shape.  What is
built: RIFF/WAVE PCM (8/16/24/32-bit integer, 32/64-bit float) parsed directly — `soundfile`, `librosa` and `pydub` are not
installed here and compressed formats (mp3 / flac / ogg) are out of scope —, integer -> float32 scaling (1 / 2^(bits-1)), channel
selection ("average", an index, a list), resampling and 30 s pad / trim.  Resampling: the reference calls
`librosa.resample` (default `soxr_hq`); soxr is absent, so `scipy.signal.resample_poly` (Kaiser-windowed polyphase FIR) is used —
**parity unpinned** against librosa (tests check length, pass-band gain and alias rejection instead).
"""
from __future__ import annotations

import math
import os
import struct
from fractions import Fraction
from typing import Iterable, List, Optional, Sequence, Union

import numpy as np
import torch

SAMPLE_RATE, N_SAMPLES = 16000, 480000


def read_wav(path_or_file) -> "tuple[np.ndarray, int]":
    """RIFF/WAVE -> (samples [n] or [n, channels] in the file's dtype, sample_rate).  PCM and IEEE float, plain or extensible."""
    f = open(path_or_file, "rb") if isinstance(path_or_file, (str, os.PathLike)) else path_or_file
    try:
        head = f.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise ValueError("not a RIFF/WAVE file")
        fmt = data = None
        while True:
            ck = f.read(8)
            if len(ck) < 8:
                break
            cid, size = ck[:4], struct.unpack("<I", ck[4:])[0]
            body = f.read(size)
            if size & 1:
                f.read(1)
            if cid == b"fmt ":
                fmt = body
            elif cid == b"data":
                data = body
                break
        if fmt is None or data is None:
            raise ValueError("WAVE file without fmt / data chunk")
        tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
        if tag == 0xFFFE and len(fmt) >= 26:                                   # WAVE_FORMAT_EXTENSIBLE: real tag in the GUID
            tag = struct.unpack("<H", fmt[24:26])[0]
        if tag == 1 and bits in (16, 32):
            x = np.frombuffer(data, dtype="<i2" if bits == 16 else "<i4")
        elif tag == 1 and bits == 8:
            x = np.frombuffer(data, dtype=np.uint8)
        elif tag == 1 and bits == 24:
            b = np.frombuffer(data[: len(data) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            x = (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)) << 8                # left-justified in int32 (full scale 2^31), like libsndfile
        elif tag == 3 and bits in (32, 64):
            x = np.frombuffer(data, dtype="<f4" if bits == 32 else "<f8")
        else:
            raise ValueError(f"unsupported WAVE format tag {tag} / {bits} bit")
        n = len(x) // ch * ch
        x = x[:n]
        return (x.reshape(-1, ch) if ch > 1 else x), sr
    finally:
        if f is not path_or_file:
            f.close()


def convert_samples_to_float32(samples: np.ndarray) -> np.ndarray:
    """Integers are scaled to [-1, 1) by 1 / 2^(bits-1), floats are cast (`audio.py` `_convert_samples_to_float32`)."""
    if isinstance(samples, torch.Tensor):
        samples = samples.detach().cpu().numpy()
    samples = np.asarray(samples)
    if samples.dtype == np.uint8:                                                  # 8-bit WAVE is offset binary
        return (samples.astype(np.float32) - 128.0) / 128.0
    if np.issubdtype(samples.dtype, np.integer):
        bits = np.iinfo(samples.dtype).bits
        return np.asarray(samples, dtype=np.float32) * np.float32(1.0 / 2 ** (bits - 1))
    if np.issubdtype(samples.dtype, np.floating):
        return np.asarray(samples, dtype=np.float32)
    raise TypeError(f"Unsupported sample type: {samples.dtype}.")


def select_channels(signal: np.ndarray, channel_selector=None) -> np.ndarray:
    """[n, channels] -> [n] / [n, k] (`audio.py:62-114`): None keeps everything, "average" is the mean over channels."""
    if signal.ndim == 1:
        if channel_selector not in (None, 0, "average"):
            raise ValueError(f"Input signal is one-dimensional, channel selector ({channel_selector}) cannot not be used.")
        return signal
    if signal.ndim > 2:
        raise NotImplementedError("Signals with more than two dimensions (sample, channel) are currently not supported.")
    nch = signal.shape[-1]
    if channel_selector is None:
        return signal
    if channel_selector == "average":
        return signal.mean(axis=-1)
    if isinstance(channel_selector, int):
        if channel_selector >= nch:
            raise ValueError(f"Cannot select channel {channel_selector} from a signal with {nch} channels.")
        return signal[..., channel_selector]
    if isinstance(channel_selector, Iterable) and not isinstance(channel_selector, str):
        sel = list(channel_selector)
        if max(sel) >= nch:
            raise ValueError(f"Cannot select channel subset {sel} from a signal with {nch} channels.")
        out = signal[..., sel]
        return out[..., 0] if len(sel) == 1 else out
    raise ValueError(f"Unexpected value for channel_selector ({channel_selector})")


def resample(samples: np.ndarray, orig_sr: int, target_sr: int) -> np.ndarray:
    """Along axis 0; output length ceil(n * target / orig) like librosa.  Polyphase Kaiser FIR (parity unpinned, see header)."""
    if orig_sr == target_sr:
        return samples
    from scipy.signal import resample_poly
    r = Fraction(int(target_sr), int(orig_sr))
    y = resample_poly(samples.astype(np.float64), r.numerator, r.denominator, axis=0, window=("kaiser", 8.6))
    n = int(math.ceil(samples.shape[0] * target_sr / orig_sr))
    if y.shape[0] < n:
        y = np.concatenate([y, np.zeros((n - y.shape[0],) + y.shape[1:], y.dtype)], axis=0)
    return y[:n].astype(np.float32)


class AudioSegment:
    """`samples` float32 [n] (or [n, channels] when no selector is given), `sample_rate`."""

    def __init__(self, samples, sample_rate: int, target_sr: Optional[int] = None, channel_selector=None, **unused):
        x = select_channels(convert_samples_to_float32(samples), channel_selector)
        if target_sr is not None and target_sr != sample_rate:
            x = resample(x, sample_rate, target_sr)
            sample_rate = target_sr
        self._samples, self._sample_rate = np.ascontiguousarray(x, dtype=np.float32), int(sample_rate)

    @classmethod
    def from_file(cls, audio_file, target_sr=None, int_values=False, offset=0, duration=0, channel_selector=None, **kw):
        """`audio_file`: a WAVE path / file object, or an already decoded (samples, sample_rate) pair / 16 kHz array."""
        if isinstance(audio_file, (np.ndarray, torch.Tensor)):
            samples, sr = audio_file, SAMPLE_RATE
        elif isinstance(audio_file, tuple):
            samples, sr = audio_file
        else:
            try:
                samples, sr = read_wav(audio_file)
            except (OSError, ValueError, struct.error) as e:
                raise Exception(f"Your audio file {audio_file} could not be decoded. We tried using the built-in WAVE reader "
                                f"(soundfile / pydub are not available): {e}") from e
        if offset > 0:
            samples = samples[int(offset * sr):]
        if duration > 0:
            samples = samples[: int(duration * sr)]
        return cls(samples, sr, target_sr=target_sr, channel_selector=channel_selector)

    @property
    def samples(self) -> np.ndarray:
        return self._samples

    @property
    def sample_rate(self) -> int:
        return self._sample_rate

    @property
    def num_samples(self) -> int:
        return self._samples.shape[0]

    @property
    def duration(self) -> float:
        return self._samples.shape[0] / float(self._sample_rate)


def pad_or_trim(waves: Sequence[Union[np.ndarray, torch.Tensor, list]], n: int = N_SAMPLES) -> torch.Tensor:
    """List of mono waveforms -> [B, n] float32, zero-padded on the right / truncated (what WhisperFeatureExtractor does with
    padding="max_length", `TF:models/whisper/feature_extraction_whisper.py:297-306`)."""
    out = torch.zeros(len(waves), n, dtype=torch.float32)
    for i, w in enumerate(waves):
        t = torch.as_tensor(np.asarray(w, dtype=np.float32) if not isinstance(w, torch.Tensor) else w.float()).reshape(-1)
        m = min(n, t.numel())
        out[i, :m] = t[:m]
    return out


class _Features:
    def __init__(self, x):
        self.input_features = x

    def __getitem__(self, k):
        return getattr(self, k)


class HipLogMelProcessor:
    """Drop-in for the `processor(...)` call of the collate function (`simple_dataset.py:239-243`): waveforms -> `[n, n_mels,
    3000]` log-mel `input_features`, computed by `desta_logmel_f32` on the device (A1) — one H2D copy of the padded clips, no
    `tolist()` round trip, no CPU STFT.  Features stay on the device; the model consumes them there."""

    def __init__(self, feature_size: int = 128, device="cuda:0"):
        self.feature_size, self.device = int(feature_size), torch.device(device)
        self.sampling_rate, self.n_samples = SAMPLE_RATE, N_SAMPLES

    def __call__(self, raw_speech, sampling_rate: Optional[int] = None, return_tensors: Optional[str] = "pt", **kw):
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(f"The model corresponding to this feature extractor was trained using a sampling rate of "
                             f"{self.sampling_rate}. Please make sure that the provided `raw_speech` input was sampled with "
                             f"{self.sampling_rate} and not {sampling_rate}.")
        from .. import _hip
        if isinstance(raw_speech, (np.ndarray, torch.Tensor)) and getattr(raw_speech, "ndim", 1) == 1:
            raw_speech = [raw_speech]
        if isinstance(raw_speech, torch.Tensor) and raw_speech.ndim == 2 and raw_speech.shape[1] == self.n_samples and raw_speech.dtype == torch.float32:
            wave = raw_speech                              # already padded / trimmed by the collate's host half (possibly in pinned memory)
        else:
            wave = pad_or_trim(raw_speech, self.n_samples)
        with torch.cuda.device(self.device):
            if wave.device.type == "cpu" and not wave.is_pinned():
                wave = wave.pin_memory()
            feats = _hip.logmel(wave.to(self.device, non_blocking=True), self.feature_size)
        return _Features(feats)
