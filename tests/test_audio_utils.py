"""CPU: decode-free audio front end (reference desta/utils/audio.py:62-166, 246-361).  soundfile / librosa are
absent, so there is no reference output to pin against: WAVE parsing is checked against python's stdlib `wave` writer and exact
integer scaling, resampling through size-independent properties (length rule, pass-band gain, alias rejection) — parity with
librosa's soxr_hq is UNPINNED and says so in the module header."""
import io
import math
import struct
import wave

import numpy as np
import pytest
import torch

from desta.utils.audio import AudioSegment, convert_samples_to_float32, pad_or_trim, read_wav, resample, select_channels


def _wav_bytes(x: np.ndarray, sr: int, width: int) -> bytes:
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(1 if x.ndim == 1 else x.shape[1])
        w.setsampwidth(width)
        w.setframerate(sr)
        w.writeframes(x.tobytes())
    return buf.getvalue()


def test_wav_pcm16_stereo_average_and_scaling(tmp_path):
    rng = np.random.default_rng(0)
    x = rng.integers(-32768, 32767, size=(1000, 2), dtype=np.int16)
    p = tmp_path / "a.wav"
    p.write_bytes(_wav_bytes(x, 16000, 2))
    raw, sr = read_wav(str(p))
    assert sr == 16000 and raw.dtype == np.int16 and np.array_equal(raw, x)
    seg = AudioSegment.from_file(str(p), target_sr=16000, channel_selector="average")
    want = (x.astype(np.float32) / 32768.0).mean(axis=1)
    assert seg.samples.dtype == np.float32 and seg.sample_rate == 16000 and seg.num_samples == 1000
    np.testing.assert_array_equal(seg.samples, want)                       # exact: int -> float32 by 2^-15, mean over channels
    assert abs(seg.duration - 1000 / 16000) < 1e-12
    np.testing.assert_array_equal(AudioSegment.from_file(str(p), channel_selector=1).samples, x[:, 1].astype(np.float32) / 32768.0)
    assert AudioSegment.from_file(str(p)).samples.shape == (1000, 2)
    assert AudioSegment.from_file(str(p), offset=0.01, duration=0.02, channel_selector=0).num_samples == 320


def test_wav_other_sample_formats(tmp_path):
    x8 = np.arange(0, 256, dtype=np.uint8)
    raw, _ = read_wav(io.BytesIO(_wav_bytes(x8, 8000, 1)))
    np.testing.assert_array_equal(convert_samples_to_float32(raw), (x8.astype(np.float32) - 128) / 128)
    # 24-bit: sign extension and 2^-23 scaling
    vals = np.array([0, 1, -1, 2 ** 23 - 1, -2 ** 23, 123456, -654321], dtype=np.int32)
    b = b"".join(struct.pack("<i", int(v))[:3] for v in vals)
    hdr = b"RIFF" + struct.pack("<I", 36 + len(b)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 44100, 44100 * 3, 3, 24) + b"data" + struct.pack("<I", len(b))
    raw, sr = read_wav(io.BytesIO(hdr + b))
    assert sr == 44100 and np.array_equal(np.asarray(raw), vals << 8)     # left-justified in int32
    np.testing.assert_array_equal(convert_samples_to_float32(raw), vals.astype(np.float32) / 2 ** 23)
    # IEEE float32
    xf = np.linspace(-1, 1, 50, dtype=np.float32)
    hdr = b"RIFF" + struct.pack("<I", 36 + xf.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 1, 16000, 64000, 4, 32) + b"data" + struct.pack("<I", xf.nbytes)
    raw, _ = read_wav(io.BytesIO(hdr + xf.tobytes()))
    np.testing.assert_array_equal(raw, xf)
    with pytest.raises(Exception, match="could not be decoded"):
        (tmp_path / "bad.wav").write_bytes(b"not a wave file at all")
        AudioSegment.from_file(str(tmp_path / "bad.wav"))
    with pytest.raises(TypeError, match="Unsupported sample type"):
        convert_samples_to_float32(np.zeros(3, dtype=np.complex64))


def test_channel_selector_errors():
    x = np.zeros((10, 3), dtype=np.float32)
    assert select_channels(x, [2]).shape == (10,) and select_channels(x, [0, 2]).shape == (10, 2)
    with pytest.raises(ValueError, match="Cannot select channel 3"):
        select_channels(x, 3)
    with pytest.raises(ValueError, match="one-dimensional"):
        select_channels(x[:, 0], 1)
    with pytest.raises(ValueError, match="Unexpected value"):
        select_channels(x, "left")


@pytest.mark.parametrize("sr", [44100, 48000, 8000, 22050])
def test_resample_properties(sr):
    n = sr * 2 + 7
    t = np.arange(n) / sr
    tone = np.sin(2 * np.pi * 440.0 * t).astype(np.float32)
    y = resample(tone, sr, 16000)
    assert y.dtype == np.float32 and y.shape[0] == math.ceil(n * 16000 / sr)          # librosa's length rule
    ref = np.sin(2 * np.pi * 440.0 * np.arange(y.shape[0]) / 16000)
    mid = slice(400, -400)
    assert np.abs(y[mid] - ref[mid]).max() < 2e-3                                      # pass band: unit gain, no delay
    if sr > 16000:                                                                     # a 10 kHz tone is above the new Nyquist
        alias = resample(np.sin(2 * np.pi * 10000.0 * t).astype(np.float32), sr, 16000)
        assert np.abs(alias[mid]).max() < 2e-2
    assert resample(tone, 16000, 16000) is tone
    # AudioSegment resamples multi-channel input along time
    seg = AudioSegment(np.stack([tone, -tone], 1), sr, target_sr=16000)
    assert seg.samples.shape == (y.shape[0], 2) and np.allclose(seg.samples[:, 0], -seg.samples[:, 1], atol=1e-6)


def test_pad_or_trim():
    out = pad_or_trim([np.ones(5, dtype=np.float32), torch.ones(12), [0.5] * 3], n=8)
    assert out.shape == (3, 8) and out.dtype == torch.float32
    assert out[0].tolist() == [1] * 5 + [0] * 3 and out[1].tolist() == [1] * 8 and out[2, :4].tolist() == [0.5, 0.5, 0.5, 0.0]
