"""Shared test helpers: oracle Dims -> product config, golden batch loading."""
import os

import torch
from safetensors.torch import load_file

import desta_oracle as O


def cfg_from_dims(d: O.Dims, dropout: float = 0.0, **extra):
    from desta.models.modeling_desta25 import DeSTA25Config
    scaling = None
    if d.rope_llama3 is not None:
        f, lo, hi, old = d.rope_llama3
        scaling = {"rope_type": "llama3", "factor": f, "low_freq_factor": lo, "high_freq_factor": hi,
                   "original_max_position_embeddings": old}
    llm = dict(model_type="qwen3" if d.qk_norm else "llama", hidden_size=d.llm_h, num_hidden_layers=d.llm_layers,
               num_attention_heads=d.llm_hq, num_key_value_heads=d.llm_hkv, head_dim=d.llm_hd,
               intermediate_size=d.llm_inter, vocab_size=d.vocab, rms_norm_eps=d.rms_eps, rope_theta=d.rope_theta,
               rope_scaling=scaling, tie_word_embeddings=d.tie_embeddings)
    enc = dict(num_mel_bins=d.n_mels, d_model=d.enc_d, encoder_layers=d.enc_layers, encoder_attention_heads=d.enc_heads,
               encoder_ffn_dim=d.enc_ffn, max_source_positions=d.enc_T)
    return DeSTA25Config(llm_model_id="local-llm", encoder_model_id="local-whisper", llm_config=llm, encoder_config=enc,
                         qformer_num_hidden_layers=d.qf_layers, prompt_size=d.prompt_size,
                         qformer_intermediate_size=d.qf_inter, target_layer_ids=list(d.taps), qformer_dropout=dropout, **extra)


def golden_batch(golden_dir, name):
    g = load_file(os.path.join(golden_dir, f"{name}.safetensors" if name.startswith("ref_") else f"ref_tiny_{name}.safetensors"))
    n = g["starts"].shape[0]
    batch = {"input_ids": g["input_ids"], "attention_mask": g["attention_mask"], "labels": g["labels"],
             "batch_features": g["batch_features"],
             "batch_start_positions": [(int(b), int(s)) for b, s in g["starts"].tolist()],
             "batch_transcription_ids": [torch.zeros(1, 0, dtype=torch.long) for _ in range(n)]}
    return g, batch


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ------------------------------------------------------------------------------------------------ toy tokenizer
class ToyTokenizer:
    """Deterministic word-level tokenizer with the slice of the HF tokenizer protocol the reference's collate / dataset code
    touches (`simple_dataset.py:130-301, 574-743`): left padding, right truncation, `return_length` = padded length, special
    tokens kept whole.  Used on BOTH sides of the collate parity check: the golden script drives the REFERENCE's BaseCollateFn
    with it, the tests drive the product's — so only the integer / index logic under test can differ."""
    padding_side = "left"
    eos_token = "<|eos|>"
    pad_token_id = 0
    eos_token_id = 2
    _SPLIT = __import__("re").compile(r"<\|[^|\s]*\|>|<[a-z_]+>|[A-Za-z0-9']+|[^\sA-Za-z0-9]")

    pad_token = None

    class _Enc(dict):
        def to(self, device):
            return self

    def add_tokens(self, toks):
        return 0

    def __init__(self, vocab_size: int = 512):
        self.vocab_size = vocab_size
        self._fixed = {"<|pad|>": 0, "<|bos|>": 1, "<|eos|>": 2, "<|AUDIO|>": 3, "<|video_pad|>": 4, "<|start|>": 5, "<|end|>": 6}

    def tokenize(self, text, add_special_tokens=False, **kw):
        return self._SPLIT.findall(text)

    def convert_tokens_to_string(self, tokens):
        return " ".join(tokens)

    def convert_tokens_to_ids(self, tokens):
        import zlib
        one = isinstance(tokens, str)
        ids = [self._fixed.get(t, 8 + zlib.crc32(t.encode()) % (self.vocab_size - 8)) for t in ([tokens] if one else tokens)]
        return ids[0] if one else ids

    def encode(self, text, add_special_tokens=False, return_tensors=None, **kw):
        ids = self.convert_tokens_to_ids(self.tokenize(text))
        return torch.tensor([ids], dtype=torch.long).reshape(1, len(ids)) if return_tensors == "pt" else ids

    def __call__(self, texts, truncation=False, padding=False, max_length=None, return_tensors=None, return_length=False,
                 add_special_tokens=False, **kw):
        rows = [self.encode(t) for t in ([texts] if isinstance(texts, str) else texts)]
        if truncation and max_length is not None:
            rows = [r[:max_length] for r in rows]                       # truncation_side = "right"
        L = max(len(r) for r in rows)
        ids = torch.full((len(rows), L), self.pad_token_id, dtype=torch.long)
        am = torch.zeros(len(rows), L, dtype=torch.long)
        for i, r in enumerate(rows):
            if r:
                ids[i, L - len(r):] = torch.tensor(r)
                am[i, L - len(r):] = 1
        out = self._Enc({"input_ids": ids, "attention_mask": am})
        if return_length:
            out["length"] = torch.full((len(rows),), L, dtype=torch.long)   # HF: length of each PADDED sequence
        return out

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=True, **kw):
        if messages and isinstance(messages[0], list):                 # a batch of conversations -> list of strings
            return [self.apply_chat_template(m, tokenize, add_generation_prompt) for m in messages]
        s = "".join(f"<|start|>{m['role']}\n{m['content']}<|end|>\n" for m in messages)
        return s + ("<|start|>assistant\n" if add_generation_prompt else "")

    def batch_decode(self, ids, skip_special_tokens=False):
        return [" ".join(str(int(t)) for t in row if not (skip_special_tokens and int(t) < 8)) for row in ids]


GENERATE_MESSAGES = [
    [{"role": "system", "content": "Focus on the audio clips and instructions."},
     {"role": "user", "content": "Hello! this is my audio <|AUDIO|>. Help me transcribe.", "audios": [{"audio": "g1.wav", "text": "hello world"}]}],
    [{"role": "user", "content": "Compare <|AUDIO|> with <|AUDIO|> please", "audios": [{"audio": "g2.wav", "text": None}, {"audio": "g3.wav", "text": ""}]}],
]

COLLATE_CASES = {
    # name: (records, batches as index lists, audio keys that fail to decode)
    "basic": ([dict(id="a.wav", prompt="Describe the audio.", response="A dog barks twice ."),
               dict(id="b.wav", prompt="What is said? <|AUDIO|> Answer briefly.", response="Hello there , general ."),
               dict(id="c.wav", prompt="Transcribe", response="one two three four five six seven eight nine ten eleven twelve"),
               dict(id="d.wav", prompt="", response="skipped: empty prompt"),
               dict(id="missing.wav", prompt="no file", response="skipped: no audio file"),
               dict(id="e.wav", prompt="Listen <start_audio>ignored text<end_audio> and answer", response="ok"),
               dict(id="f.wav", prompt="empty response is skipped", response="")],
              [[0, 1, 2], [1], [2, 3, 0], [3], [2]], {"c.wav"}),
}


# ------------------------------------------------------------------------------------------------ Whisper decoder stand-in (ASR leg of generate)
ASR_DIMS = dict(decoder_layers=2, vocab_size=96, max_target_positions=64)
ASR_GEN_CFG = dict(decoder_start_token_id=1, eos_token_id=2, pad_token_id=2, bos_token_id=2, max_length=64, is_multilingual=True,
                   lang_to_id={"<|en|>": 10, "<|de|>": 11, "<|fr|>": 12}, task_to_id={"transcribe": 13, "translate": 14}, no_timestamps_token_id=15,
                   suppress_tokens=[5, 6, 20], begin_suppress_tokens=[7, 2], return_timestamps=False)


def asr_weights(d: O.Dims, seed: int = 5):
    """Seeded Whisper DECODER weights (+ the encoder's final layer_norm, which the perception taps skip) at tiny width under the
    checkpoint's names — regenerated identically by the golden script and by the GPU test, never stored."""
    g = torch.Generator().manual_seed(seed)
    w, dm, ffn, V, P = {}, d.enc_d, d.enc_ffn, ASR_DIMS["vocab_size"], ASR_DIMS["max_target_positions"]
    E, D = "perception.whisper.model.encoder.", "perception.whisper.model.decoder."

    def lin(name, o, i, bias=True):
        w[name + ".weight"] = 0.3 * torch.randn(o, i, generator=g)            # (large on purpose: a random decoder with small weights repeats one token)
        if bias:
            w[name + ".bias"] = 0.1 * torch.randn(o, generator=g)

    def ln(name):
        w[name + ".weight"] = 1.0 + 0.1 * torch.randn(dm, generator=g)
        w[name + ".bias"] = 0.1 * torch.randn(dm, generator=g)
    ln(E + "layer_norm")
    w[D + "embed_tokens.weight"] = torch.randn(V, dm, generator=g)
    w[D + "embed_positions.weight"] = 0.3 * torch.randn(P, dm, generator=g)
    for i in range(ASR_DIMS["decoder_layers"]):
        p = f"{D}layers.{i}."
        for a in ("self_attn", "encoder_attn"):
            lin(p + a + ".q_proj", dm, dm)
            lin(p + a + ".k_proj", dm, dm, bias=False)
            lin(p + a + ".v_proj", dm, dm)
            lin(p + a + ".out_proj", dm, dm)
            ln(p + a + "_layer_norm")
        lin(p + "fc1", ffn, dm)
        lin(p + "fc2", dm, ffn)
        ln(p + "final_layer_norm")
    ln(D + "layer_norm")
    return w
