"""ctypes binding of libdesta_hip.so (the C ABI declared in include/desta_hip.h).

PyTorch is used for device memory and streams only; every op below enqueues hand-written HIP
kernels on torch's CURRENT stream through raw pointers.  There is no fallback: if the library is
missing, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: its libamdhip64.so.7 is the runtime we bind to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DESTA_HIP_LIB") or os.path.join(_HERE, "lib", "libdesta_hip.so")     # env override: kernel A/B builds

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python desta2.5-audio_amd/build.py` "
        "(hipcc --offload-arch=gfx950). There is no CPU/eager fallback for the DeSTA2.5 hot path.")

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp),
                ("M", i32), ("N", i32), ("K", i32), ("batch", i32),
                ("lda", i64), ("ldb", i64), ("ldc", i64),
                ("stride_a", i64), ("stride_b", i64), ("stride_c", i64),
                ("bias", vp), ("residual", vp), ("ldr", i64), ("stride_r", i64), ("residual_f32", i32),
                ("act", i32), ("out_f32", i32),
                ("preact", vp), ("ldp", i64), ("stride_p", i64), ("alpha", f32),
                ("aux", vp), ("ld_aux", i64), ("dropout_p", f32), ("dropout_seed", C.c_uint64),
                ("workspace", vp), ("workspace_bytes", C.c_size_t),
                ("trans_a", i32), ("trans_b", i32), ("a_rms_weight", vp), ("a_rms_eps", f32),
                ("rope_cos_sin", vp), ("rope_pos", vp), ("rope_cols", i32), ("rope_head_dim", i32)]


lib.desta_abi_version.restype = i32
lib.desta_last_error.restype = C.c_char_p
ABI_VERSION = 7
if lib.desta_abi_version() != ABI_VERSION:
    raise ImportError(f"libdesta_hip.so has ABI version {lib.desta_abi_version()}, this binding needs {ABI_VERSION}: "
                      "rebuild with `python desta2.5-audio_amd/build.py`")


def _sig(name, *argtypes):
    fn = getattr(lib, name)
    fn.argtypes = list(argtypes)
    fn.restype = i32
    return fn


def check(ret: int, what: str) -> None:
    if ret != 0:
        raise RuntimeError(f"{what} failed ({ret}): {lib.desta_last_error().decode()}")


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def p(t) -> int:
    """Device pointer of a tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


_gemm = _sig("desta_gemm_bf16_nt", C.POINTER(GemmDesc), vp)


def gemm(A, B, out, M, N, K, *, lda=None, ldb=None, ldc=None, bias=None, residual=None, ldr=None,
         act=0, preact=None, ldp=None, alpha=1.0, batch=1, stride_a=0, stride_b=0, stride_c=0,
         stride_r=0, stride_p=0, aux=None, ld_aux=0, dropout_p=0.0, dropout_seed=0, a_rms_weight=None, a_rms_eps=0.0,
         trans_a=False, trans_b=False, rope=None):
    """out[M,N] = act(alpha * A[M,K] @ B[N,K]^T + bias) + residual  (bf16 operands, MFMA).
    a_rms_weight: decode path, RMSNorm(A; weight, eps) fused into the projection (see `rms_fusable`).
    rope = (cos_sin [positions, hd/2, 2] fp32, pos [M] int32, cols, head_dim): rotary embedding of output columns [0, cols) in
    adjacent pairs inside the epilogue (include/desta_hip.h, desta_gemm_desc.rope_*)."""
    d = GemmDesc()
    d.A, d.B, d.C = p(A), p(B), p(out)
    d.M, d.N, d.K, d.batch = M, N, K, batch
    d.lda = (M if trans_a else K) if lda is None else lda       # transposed storage: A is [K, M], B is [K, N]
    d.ldb = (N if trans_b else K) if ldb is None else ldb
    d.trans_a, d.trans_b = int(trans_a), int(trans_b)
    d.ldc = N if ldc is None else ldc
    d.stride_a, d.stride_b, d.stride_c = stride_a, stride_b, stride_c
    d.bias = p(bias)
    d.residual = p(residual)
    d.ldr = (N if ldr is None else ldr)
    d.stride_r = stride_r
    d.residual_f32 = int(residual is not None and residual.dtype == torch.float32)
    d.act = act
    d.out_f32 = int(out.dtype == torch.float32)
    d.preact = p(preact)
    d.ldp = N if ldp is None else ldp
    d.stride_p = stride_p
    d.alpha = alpha
    d.aux, d.ld_aux = p(aux), ld_aux
    d.dropout_p, d.dropout_seed = dropout_p, dropout_seed
    d.a_rms_weight, d.a_rms_eps = p(a_rms_weight), a_rms_eps
    if rope is not None:
        cs_, pos_, cols_, hd_ = rope
        assert cs_.dtype == torch.float32 and pos_.dtype == torch.int32 and pos_.numel() >= M and cs_.is_contiguous()
        d.rope_cos_sin, d.rope_pos, d.rope_cols, d.rope_head_dim = p(cs_), p(pos_), int(cols_), int(hd_)
    st = stream()
    ws = _gemm_ws.get((A.device, st))                       # one split-K scratch per STREAM: GEMMs of one stream run serially,
    if ws is None:                                          # GEMMs of two streams (encoder prefetch beside the LLM) must not share it
        ws = _gemm_ws[(A.device, st)] = torch.zeros(GEMM_WS_BYTES // 4, dtype=torch.float32, device=A.device)   # zeroed: the arrival tickets of the in-kernel split-K reduce live in its last 4 KiB
    d.workspace, d.workspace_bytes = ws.data_ptr(), GEMM_WS_BYTES
    if _gemm_prof is not None:
        # HIP events on the launch stream around this one kernel (bench.py roofline leg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_gemm(C.byref(d), st), "desta_gemm_bf16_nt")
        e1.record()
        _gemm_prof.append((e0, e1, 2.0 * M * N * K * batch, lib.desta_gemm_last_kernel()))
        return out
    check(_gemm(C.byref(d), st), "desta_gemm_bf16_nt")
    return out


def rms_fusable(M: int, K: int) -> bool:
    """True when `gemm(..., a_rms_weight=)` can normalise the M rows of A in LDS (decode path)."""
    return M <= 16 and M * 2 * K <= 8 * 4096 * 2 and K % 512 == 0


_gemm_prof = None
_gemm_ws = {}                      # (device, stream) -> split-K scratch
GEMM_WS_BYTES = (64 << 20) + 4096


def gemm_profile_start():
    global _gemm_prof
    _gemm_prof = []


def gemm_profile_stop(by_kernel: bool = False):
    """-> (n_launches, total_flops, total_ms) of the GEMM launches since gemm_profile_start(); with by_kernel a dict
    {kernel family (1 = 128x128, 2 = 256x256, 3 = skinny): (n, flops, ms)} instead."""
    global _gemm_prof
    rec, _gemm_prof = _gemm_prof, None
    torch.cuda.synchronize()
    if by_kernel:
        out = {}
        for a, b, f, k in rec:
            n0, f0, m0 = out.get(k, (0, 0.0, 0.0))
            out[k] = (n0 + 1, f0 + f, m0 + a.elapsed_time(b))
        return out
    ms = sum(a.elapsed_time(b) for a, b, _, _ in rec)
    return len(rec), sum(f for _, _, f, _ in rec), ms


# HBM-bound / attention kernels: HIP events around the launches of a tagged wrapper while a profile is open (bench.py's
# `roofline.hbm_kernels` leg; never on in the timed region).  `work` = ALGORITHMIC bytes (or FLOP for attention) of the call.
_kprof = None


def kernel_profile_start():
    global _kprof
    _kprof = {}


def kernel_profile_stop():
    """-> {tag: (n_calls, total_work, total_ms)}"""
    global _kprof
    rec, _kprof = _kprof, None
    torch.cuda.synchronize()
    return {tag: (len(v), sum(w for _, _, w in v), sum(a.elapsed_time(b) for a, b, _ in v)) for tag, v in rec.items()}


def _profiled(tag, work):
    """Decorator: `work(*args, **kw)` -> algorithmic bytes (FLOP) of the call; evaluated only while profiling."""
    def deco(fn):
        def wrapped(*a, **kw):
            if _kprof is None:
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            _kprof.setdefault(tag if isinstance(tag, str) else tag(*a, **kw), []).append((e0, e1, float(work(*a, **kw))))
            return out
        wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
        return wrapped
    return deco


# ----------------------------------------------------------------------------- log-mel
lib.desta_logmel_table_floats.restype = C.c_size_t
lib.desta_logmel_table_floats.argtypes = [i32]
lib.desta_logmel_workspace_floats.restype = C.c_size_t
lib.desta_logmel_workspace_floats.argtypes = [i32]
_logmel_fill = _sig("desta_logmel_fill_tables", i32, vp)
_logmel = _sig("desta_logmel_f32", vp, i32, i32, i64, vp, i32, vp, vp, vp)

_logmel_tables = {}


def logmel_tables(n_mels: int, device) -> torch.Tensor:
    key = (n_mels, str(device))
    if key not in _logmel_tables:
        n = lib.desta_logmel_table_floats(n_mels)
        host = torch.empty(n, dtype=torch.float32)
        check(_logmel_fill(n_mels, host.data_ptr()), "desta_logmel_fill_tables")
        _logmel_tables[key] = host.to(device)
    return _logmel_tables[key]


@_profiled("logmel", lambda wave, n_mels, out=None: wave.shape[0] * (4 * min(wave.shape[1], 480000) + 4 * n_mels * 3000))
def logmel(wave: torch.Tensor, n_mels: int, out: torch.Tensor = None) -> torch.Tensor:
    """wave [B, n] f32 (cuda) -> [B, n_mels, 3000] f32; 30-s zero-pad/truncate semantics."""
    assert wave.is_cuda and wave.dtype == torch.float32 and wave.dim() == 2 and wave.stride(1) == 1
    B, n = wave.shape
    if out is None:
        out = torch.empty(B, n_mels, 3000, dtype=torch.float32, device=wave.device)
    ws = torch.empty(lib.desta_logmel_workspace_floats(B), dtype=torch.float32, device=wave.device)
    tb = logmel_tables(n_mels, wave.device)
    check(_logmel(p(wave), B, n, wave.stride(0), p(tb), n_mels, p(out), p(ws), stream()), "desta_logmel_f32")
    return out


# ----------------------------------------------------------------------------- optimizer
class OptPlan(C.Structure):
    _fields_ = [("tensors", vp), ("tensor_wd", vp), ("n_tensors", i32),
                ("units", vp), ("unit_col_off", vp), ("n_units", i32),
                ("vecs", vp), ("vec_wd", vp), ("n_vec", i32),
                ("sum_rows", i64), ("sum_cols", i64), ("max_batch", i32), ("max_cols", i32),
                ("chunks", vp), ("ten_chunks", vp), ("n_chunks", i32), ("max_chunks_per_tensor", i32),
                ("fin", vp), ("n_fin", i32), ("colpart_floats", i64), ("cols_multiple_of_4", i32),
                ("group_bounds", vp), ("n_groups", i32), ("ragged_units", vp), ("n_ragged", i32)]


lib.desta_adafactor_workspace_floats.restype = C.c_size_t
lib.desta_adafactor_workspace_floats.argtypes = [i32, i32, i64, i64, i64]
lib.desta_adafactor_workspace_floats_v3.restype = C.c_size_t
lib.desta_adafactor_workspace_floats_v3.argtypes = [C.POINTER(OptPlan), i64]
_adafactor = _sig("desta_clip_adafactor_step", C.POINTER(OptPlan), vp, vp, vp, vp, f32, f32, f32, f32, f32, vp)


@_profiled("clip_adafactor", lambda plan, params, *a, **k: 12 * params.numel())          # read g, read p, write p (fp32)
def clip_adafactor_step(plan: OptPlan, params, grads, state, workspace, lr, beta2t, eps1, clip_threshold,
                        max_grad_norm):
    check(_adafactor(C.byref(plan), p(params), p(grads), p(state), p(workspace), lr, beta2t, eps1,
                     clip_threshold, max_grad_norm, stream()), "desta_clip_adafactor_step")


# ----------------------------------------------------------------------------- norms / activations / layout
c_size_t = C.c_size_t
_ln_fwd = _sig("desta_layernorm_fwd", vp, i32, vp, vp, f32, i32, i32, vp, vp, vp, vp)
lib.desta_layernorm_bwd_workspace_floats.restype = c_size_t
lib.desta_layernorm_bwd_workspace_floats.argtypes = [i32, i32]
_ln_bwd = _sig("desta_layernorm_bwd", vp, i32, vp, i32, vp, vp, i32, i32, vp, vp, vp, vp, i32, vp, vp)
_rms_fwd = _sig("desta_rmsnorm_fwd", vp, vp, f32, i32, i32, vp, vp, vp)
_rms_bwd = _sig("desta_rmsnorm_bwd", vp, vp, vp, vp, vp, i32, i32, vp, vp)
lib.desta_colsum_workspace_floats.restype = c_size_t
lib.desta_colsum_workspace_floats.argtypes = [i32, i32]
_colsum = _sig("desta_colsum_bf16", vp, i32, i32, i64, vp, i32, vp, vp)
_rope = _sig("desta_rope", vp, i64, i32, i32, i32, i32, i32, vp, vp, vp, f32, vp, i64, i32, vp, i32, vp)
_swiglu_fwd = _sig("desta_swiglu_fwd", vp, vp, i64, i32, vp)
_swiglu_bwd = _sig("desta_swiglu_bwd", vp, vp, vp, i64, i32, vp)
_gelu_bwd = _sig("desta_gelu_bwd", vp, vp, vp, i64, vp)
_cast = _sig("desta_cast_f32_bf16", vp, vp, i64, vp)
_add = _sig("desta_add_f32", vp, vp, i64, vp)
_transpose = _sig("desta_transpose_to_bf16", vp, i32, i64, i32, i32, vp, i64, vp)
_mel_rows = _sig("desta_mel_to_rows", vp, i32, i32, i32, i32, vp, vp)

_scratch = {}


def scratch(nfloats: int, device, tag="ws") -> torch.Tensor:
    """Grow-only fp32 scratch buffer per (device, tag); callers on one stream reuse it serially."""
    key = (str(device), tag)
    t = _scratch.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.empty(max(nfloats, 1024), dtype=torch.float32, device=device)
        _scratch[key] = t
    return t


@_profiled("layernorm_fwd", lambda x, g, b, eps, y16=None, y32=None, stats=None: x.numel() * (x.element_size() + (2 if y16 is not None else 0) + (4 if y32 is not None else 0)))
def layernorm_fwd(x, gamma, beta, eps, y16=None, y32=None, stats=None):
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    check(_ln_fwd(p(x), int(x.dtype == torch.float32), p(gamma), p(beta), eps, rows, cols, p(y16), p(y32), p(stats),
                  stream()), "desta_layernorm_fwd")


@_profiled("layernorm_bwd", lambda dy, x, g, st, dx32=None, dx16=None, **k: x.numel() * (dy.element_size() + x.element_size() + (4 if dx32 is not None else 0) + (2 if dx16 is not None else 0)))
def layernorm_bwd(dy, x, gamma, stats, dx32=None, dx16=None, dgamma=None, dbeta=None, accumulate=False):
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    ws = scratch(lib.desta_layernorm_bwd_workspace_floats(rows, cols), x.device) if dgamma is not None else None
    check(_ln_bwd(p(dy), int(dy.dtype == torch.float32), p(x), int(x.dtype == torch.float32), p(gamma), p(stats), rows,
                  cols, p(dx32), p(dx16), p(dgamma), p(dbeta), int(accumulate), p(ws), stream()), "desta_layernorm_bwd")


@_profiled("rmsnorm_fwd", lambda x, w, eps, y, rstd=None: 4 * x.numel())                       # bf16 in + bf16 out
def rmsnorm_fwd(x, weight, eps, y, rstd=None):
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    check(_rms_fwd(p(x), p(weight), eps, rows, cols, p(y), p(rstd), stream()), "desta_rmsnorm_fwd")


@_profiled("rmsnorm_bwd", lambda dy, x, w, rstd, dx, dres=None: x.numel() * (6 + (2 if dres is not None else 0)))
def rmsnorm_bwd(dy, x, weight, rstd, dx, dres=None):
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    check(_rms_bwd(p(dy), p(x), p(weight), p(rstd), p(dres), rows, cols, p(dx), stream()), "desta_rmsnorm_bwd")


def colsum(x, rows, cols, ld, out, accumulate=False, tag="ws"):
    ws = scratch(lib.desta_colsum_workspace_floats(rows, cols), x.device, tag)     # one scratch tag per stream
    check(_colsum(p(x), rows, cols, ld, p(out), int(accumulate), p(ws), stream()), "desta_colsum_bf16")


@_profiled("rope", lambda buf, ld, rows, seq, n_q, n_kv, hd, *a, **k: 4 * rows * (n_q + n_kv) * hd)      # q|k read + written in place
def rope(buf, ld, rows, seq, n_q, n_kv, hd, cos_sin, q_norm_w=None, k_norm_w=None, eps=1e-6, pre_norm=None,
         ld_pre=0, backward=False, pos_shift=None, s_major_batch=0):
    check(_rope(p(buf), ld, rows, seq, n_q, n_kv, hd, p(cos_sin), p(q_norm_w), p(k_norm_w), eps, p(pre_norm), ld_pre,
                int(backward), p(pos_shift), s_major_batch, stream()), "desta_rope")


@_profiled("swiglu_fwd", lambda gu, act, rows, inter: 6 * rows * inter)                    # gate|up read, act written (bf16)
def swiglu_fwd(gate_up, act, rows, inter):
    check(_swiglu_fwd(p(gate_up), p(act), rows, inter, stream()), "desta_swiglu_fwd")


@_profiled("swiglu_bwd", lambda gu, dact, dgu, rows, inter: 10 * rows * inter)            # gate|up + d_act read, d(gate|up) written
def swiglu_bwd(gate_up, dact, dgate_up, rows, inter):
    check(_swiglu_bwd(p(gate_up), p(dact), p(dgate_up), rows, inter, stream()), "desta_swiglu_bwd")


def gelu_bwd(preact, dact, dpre, n):
    check(_gelu_bwd(p(preact), p(dact), p(dpre), n, stream()), "desta_gelu_bwd")


def cast_bf16(x, y, n=None):
    check(_cast(p(x), p(y), x.numel() if n is None else n, stream()), "desta_cast_f32_bf16")


def add_f32(y, x, n=None):
    check(_add(p(y), p(x), y.numel() if n is None else n, stream()), "desta_add_f32")


def transpose_to_bf16(src, rows, cols, out, ld_out, ld_in=None):
    check(_transpose(p(src), int(src.dtype == torch.float32), cols if ld_in is None else ld_in, rows, cols, p(out),
                     ld_out, stream()), "desta_transpose_to_bf16")


def mel_to_rows(mel, c_pad, out):
    B, Cn, T = mel.shape
    check(_mel_rows(p(mel), B, Cn, T, c_pad, p(out), stream()), "desta_mel_to_rows")


# ----------------------------------------------------------------------------- embed / CE / tap mix
_embed = _sig("desta_embed_gather", vp, vp, vp, i32, i32, vp, vp)
_gather = _sig("desta_gather_rows_bf16", vp, vp, i32, i32, vp, vp)
lib.desta_ce_workspace_floats.restype = c_size_t
lib.desta_ce_workspace_floats.argtypes = [i32, i32]
_ce = _sig("desta_causal_lm_loss", vp, i64, vp, i32, i32, i32, vp, vp, i32, vp)
_mix_fwd = _sig("desta_tap_mix_fwd", vp, vp, i32, i32, i32, i32, vp, vp)
_mix_bwd = _sig("desta_tap_mix_bwd", vp, vp, vp, i32, i32, i32, i32, vp, vp, vp)


def embed_gather(table, audio_rows, src_row, rows, hidden, out):
    check(_embed(p(table), p(audio_rows), p(src_row), rows, hidden, p(out), stream()), "desta_embed_gather")


def gather_rows(src, idx, rows, hidden, out):
    check(_gather(p(src), p(idx), rows, hidden, p(out), stream()), "desta_gather_rows_bf16")


@_profiled("causal_lm_loss", lambda logits, ld, labels, batch, seq, vocab, loss, write_grad=True: (2 + 2 * int(write_grad)) * batch * (seq - 1) * vocab)
def causal_lm_loss(logits, ld, labels, batch, seq, vocab, loss, write_grad=True):
    ws = scratch(lib.desta_ce_workspace_floats(batch, seq), logits.device, "ce")
    check(_ce(p(logits), ld, p(labels), batch, seq, vocab, p(loss), p(ws), int(write_grad), stream()),
          "desta_causal_lm_loss")


def tap_mix_fwd(x, lw, taps, batch, prompt, d, out):
    check(_mix_fwd(p(x), p(lw), taps, batch, prompt, d, p(out), stream()), "desta_tap_mix_fwd")


def tap_mix_bwd(x, lw, dout, taps, batch, prompt, d, dx, dlw):
    check(_mix_bwd(p(x), p(lw), p(dout), taps, batch, prompt, d, p(dx), p(dlw), stream()), "desta_tap_mix_bwd")


# ----------------------------------------------------------------------------- attention
class AttnDesc(C.Structure):
    _fields_ = [("Q", vp), ("K", vp), ("V", vp), ("O", vp), ("dO", vp), ("dQ", vp), ("dK", vp), ("dV", vp),
                ("lse", vp),
                ("q_batch_stride", i64), ("q_row_stride", i64), ("k_batch_stride", i64), ("k_row_stride", i64),
                ("v_batch_stride", i64), ("v_row_stride", i64), ("o_batch_stride", i64), ("o_row_stride", i64),
                ("do_batch_stride", i64), ("do_row_stride", i64), ("dq_batch_stride", i64), ("dq_row_stride", i64),
                ("dk_batch_stride", i64), ("dk_row_stride", i64), ("dv_batch_stride", i64), ("dv_row_stride", i64),
                ("batch", i32), ("n_q_heads", i32), ("n_kv_heads", i32), ("seq_q", i32), ("seq_k", i32),
                ("head_dim", i32), ("causal", i32), ("kv_start", vp), ("scale", f32),
                ("dropout_p", f32), ("dropout_seed", C.c_uint64), ("rope_cos_sin", vp), ("O_f32", vp),
                ("dkv_transposed", i32), ("dkv_t_ld", i64), ("dkv_bias_grad", vp)]


_attn_fwd = _sig("desta_attention_fwd", C.POINTER(AttnDesc), vp)
lib.desta_attention_bwd_workspace_floats.restype = c_size_t
lib.desta_attention_bwd_workspace_floats.argtypes = [i32, i32, i32]
_attn_bwd = _sig("desta_attention_bwd", C.POINTER(AttnDesc), vp, vp)


def _elem_ptr(t, offset_elems):
    return t.data_ptr() + offset_elems * t.element_size()


def attn_desc(q, k, v, o, lse, *, batch, hq, hkv, sq, sk, hd, scale, causal=False, kv_start=None,
              q_off=0, k_off=0, v_off=0, q_rs=None, k_rs=None, v_rs=None, o_rs=None, dropout_p=0.0, dropout_seed=0,
              q_bs=None, k_bs=None, v_bs=None, o_bs=None, o_f32=None):
    """q/k/v are 2-D row-major [batch*seq, row_stride] buffers (possibly the same fused buffer);
    *_off = first column of the q/k/v slice.  `o_f32`: optional fp32 buffer of O's shape; forward also writes the unrounded
    output there and backward takes delta = rowsum(dO * O) from it (see include/desta_hip.h)."""
    d = AttnDesc()
    q_rs = q.shape[-1] if q_rs is None else q_rs
    k_rs = k.shape[-1] if k_rs is None else k_rs
    v_rs = v.shape[-1] if v_rs is None else v_rs
    o_rs = o.shape[-1] if o_rs is None else o_rs
    d.Q, d.K, d.V, d.O = _elem_ptr(q, q_off), _elem_ptr(k, k_off), _elem_ptr(v, v_off), p(o)
    d.lse = p(lse)
    d.q_row_stride, d.q_batch_stride = q_rs, (sq * q_rs if q_bs is None else q_bs)
    d.k_row_stride, d.k_batch_stride = k_rs, (sk * k_rs if k_bs is None else k_bs)     # explicit batch strides: KV cache
    d.v_row_stride, d.v_batch_stride = v_rs, (sk * v_rs if v_bs is None else v_bs)
    d.o_row_stride, d.o_batch_stride = o_rs, (sq * o_rs if o_bs is None else o_bs)
    d.batch, d.n_q_heads, d.n_kv_heads, d.seq_q, d.seq_k, d.head_dim = batch, hq, hkv, sq, sk, hd
    d.causal = int(causal)
    d.kv_start = p(kv_start)
    d.scale = scale
    d.dropout_p, d.dropout_seed = dropout_p, dropout_seed
    if o_f32 is not None:
        assert o_f32.dtype == torch.float32 and o_f32.shape == o.shape and o_f32.is_contiguous() and o.is_contiguous()
    d.O_f32 = p(o_f32)
    d._keep = (q, k, v, o, lse, kv_start, o_f32)       # the descriptor holds RAW pointers: keep the tensors alive with it
    return d


def _attn_flops(d, mult):
    """Executed MFMA FLOP: 4*Sq*Sk*D per (batch, head) forward (QK^T + PV), halved under the causal mask."""
    f = 4.0 * d.batch * d.n_q_heads * d.seq_q * d.seq_k * d.head_dim
    return mult * (f * 0.5 if d.causal else f)


def _attn_tag(d, *a, **k):
    return f"D{d.head_dim}{'c' if d.causal else ''}_q{d.seq_q}_k{d.seq_k}"


@_profiled(lambda d: "attn_fwd:" + _attn_tag(d), lambda d: _attn_flops(d, 1.0))
def attention_fwd(d: AttnDesc):
    check(_attn_fwd(C.byref(d), stream()), "desta_attention_fwd")


@_profiled(lambda d, *a, **k: "attn_bwd:" + _attn_tag(d), lambda d, *a, **k: _attn_flops(d, 2.5))     # dS, dP recompute, dQ, dK, dV
def attention_bwd(d: AttnDesc, do, dq, dk=None, dv=None, *, do_rs=None, dq_off=0, dk_off=0, dv_off=0, dq_rs=None,
                  dk_rs=None, dv_rs=None, do_bs=None, dq_bs=None, dk_bs=None, dv_bs=None, rope_cos_sin=None, dkv_t=None):
    """Backward of the attention described by `d` (O and lse filled by forward); *_bs override the batch strides
    (position-major token grids: row stride = batch * width, batch stride = width).  rope_cos_sin [seq, hd/2, 2] fp32: Q / K
    are rotary-embedded projections in the adjacent-pair layout; dQ / dK are rotated back before the store."""
    d.rope_cos_sin = p(rope_cos_sin)
    d._keep_rope = rope_cos_sin
    do_rs = do.shape[-1] if do_rs is None else do_rs
    dq_rs = dq.shape[-1] if dq_rs is None else dq_rs
    d.dO, d.do_row_stride, d.do_batch_stride = p(do), do_rs, (d.seq_q * do_rs if do_bs is None else do_bs)
    d.dQ, d.dq_row_stride, d.dq_batch_stride = _elem_ptr(dq, dq_off), dq_rs, (d.seq_q * dq_rs if dq_bs is None else dq_bs)
    d.dkv_transposed, d.dkv_t_ld, d.dkv_bias_grad = 0, 0, 0
    if dkv_t is not None:
        # one-query-tile path: dK | dV written transposed into `t` [2 * heads * 64, ld] (K rows first), bias gradients into `bias` [2 * heads * 64] fp32
        t, ld, bias = dkv_t
        assert dk is None and dv is None and t.dtype == torch.bfloat16 and (bias is None or bias.dtype == torch.float32)
        d.dK, d.dV = p(t), _elem_ptr(t, d.n_q_heads * d.head_dim * ld)
        d.dkv_transposed, d.dkv_t_ld, d.dkv_bias_grad = 1, ld, p(bias)
        dk = dv = t
    elif dk is not None:
        dk_rs = dk.shape[-1] if dk_rs is None else dk_rs
        dv_rs = dv.shape[-1] if dv_rs is None else dv_rs
        d.dK, d.dk_row_stride, d.dk_batch_stride = _elem_ptr(dk, dk_off), dk_rs, (d.seq_k * dk_rs if dk_bs is None else dk_bs)
        d.dV, d.dv_row_stride, d.dv_batch_stride = _elem_ptr(dv, dv_off), dv_rs, (d.seq_k * dv_rs if dv_bs is None else dv_bs)
    else:
        d.dK = d.dV = 0
    ws = scratch(lib.desta_attention_bwd_workspace_floats(d.batch, d.n_q_heads, d.seq_q), do.device, "attn")
    d._keep_bwd = (do, dq, dk, dv, ws)
    check(_attn_bwd(C.byref(d), p(ws), stream()), "desta_attention_bwd")


_prompt_expand = _sig("desta_prompt_expand", vp, i32, i32, i64, vp, vp, vp)
_prompt_grad = _sig("desta_prompt_grad", vp, i32, i32, i64, vp, vp)


def prompt_expand(prompts, taps, batch, n, x32, x16):
    check(_prompt_expand(p(prompts), taps, batch, n, p(x32), p(x16), stream()), "desta_prompt_expand")


def prompt_grad(dx, taps, batch, n, dprompts):
    check(_prompt_grad(p(dx), taps, batch, n, p(dprompts), stream()), "desta_prompt_grad")


def gemm_force_variant(v: int) -> None:
    """Tuning / tests: 0 = automatic tile choice, 1 = 128x128 kernel, 2 = 256x256 8-phase kernel."""
    lib.desta_gemm_force_variant.argtypes = [i32]
    lib.desta_gemm_force_variant(v)


_dropout = _sig("desta_dropout_bf16", vp, vp, i32, i32, i64, f32, C.c_uint64, vp)
_dropout_mask = _sig("desta_dropout_mask_u8", C.c_uint64, i64, f32, vp, vp)


def dropout_bf16(x, y, rows, cols, ld, p_drop, seed):
    check(_dropout(p(x), p(y), rows, cols, ld, p_drop, seed, stream()), "desta_dropout_bf16")


def dropout_mask(seed, n, p_drop, device="cuda"):
    out = torch.empty(n, dtype=torch.uint8, device=device)
    check(_dropout_mask(seed, n, p_drop, p(out), stream()), "desta_dropout_mask_u8")
    return out


def gemm_set_persistent(on: bool) -> None:
    lib.desta_gemm_set_persistent.argtypes = [i32]
    lib.desta_gemm_set_persistent(int(on))


def gemm_set_option(option: int, value: int) -> None:
    """A/B switches of the automatic GEMM choice: 0 = persistent kernel, 1 = staggered schedule."""
    lib.desta_gemm_set_option.argtypes = [i32, i32]
    check(lib.desta_gemm_set_option(option, value), "desta_gemm_set_option")


_argmax = _sig("desta_argmax_bf16", vp, i64, i32, i32, vp, vp, vp)
lib.desta_argmax_workspace_bytes.restype = C.c_size_t
lib.desta_argmax_workspace_bytes.argtypes = [i32]
_argmax_ws = {}


def argmax_bf16(x, ld, rows, cols, out):
    key = (x.device, rows)
    ws = _argmax_ws.get(key)
    if ws is None:
        ws = _argmax_ws[key] = torch.empty(lib.desta_argmax_workspace_bytes(rows), dtype=torch.uint8, device=x.device)
    check(_argmax(p(x), ld, rows, cols, p(out), p(ws), stream()), "desta_argmax_bf16")


_mask_tokens = _sig("desta_mask_tokens_bf16", vp, i64, i32, i32, vp, i32, vp)


def mask_tokens_bf16(logits, ld, rows, cols, ids):
    """logits[:, ids] = -inf (ids int32 on the device)."""
    check(_mask_tokens(p(logits), ld, rows, cols, p(ids), int(ids.numel()), stream()), "desta_mask_tokens_bf16")


_rope_kv = _sig("desta_rope_kv_append", vp, i64, i32, i32, i32, i32, i32, vp, vp, vp, f32, vp, vp, i64, i64, i32, vp)


def rope_kv_append(buf, ld, rows, seq, n_q, n_kv, hd, cos_sin, q_norm_w, k_norm_w, eps, pos_shift, cache, kv_bs, kv_rs, slot0):
    """Forward rope on a fused q|k|v buffer + append of the rotated K and the V heads to the KV cache slab."""
    check(_rope_kv(p(buf), ld, rows, seq, n_q, n_kv, hd, p(cos_sin), p(q_norm_w), p(k_norm_w), eps, p(pos_shift), p(cache), kv_bs,
                   kv_rs, slot0, stream()), "desta_rope_kv_append")


_sample = _sig("desta_sample_top_p_bf16", vp, i64, i32, i32, f32, f32, C.c_uint64, C.c_uint32, vp, vp, vp)


def sample_top_p(logits, ld, rows, cols, temperature, top_p, seed, step, out, keep_mask=None):
    """One temperature / top-p sample per row of bf16 logits (HF `generate(do_sample=True)` step)."""
    check(_sample(p(logits), ld, rows, cols, temperature, top_p, seed & 0xFFFFFFFFFFFFFFFF, step & 0xFFFFFFFF, p(out), p(keep_mask),
                  stream()), "desta_sample_top_p_bf16")


# ----------------------------------------------------------------------------- data-parallel exchange without torch.distributed
class CommUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def comm_get_unique_id() -> bytes:
    """128 opaque bytes (ncclUniqueId) made on rank 0; every rank passes them to `comm_create`."""
    lib.desta_comm_get_unique_id.argtypes = [C.POINTER(CommUniqueId)]
    uid = CommUniqueId()
    check(lib.desta_comm_get_unique_id(C.byref(uid)), "desta_comm_get_unique_id")
    return bytes(C.string_at(C.addressof(uid), 128))


def comm_create(world_size: int, rank: int, unique_id: bytes) -> int:
    lib.desta_comm_create.argtypes = [C.POINTER(vp), i32, i32, C.POINTER(CommUniqueId)]
    uid = CommUniqueId()
    C.memmove(C.addressof(uid), unique_id, 128)
    comm = vp()
    check(lib.desta_comm_create(C.byref(comm), world_size, rank, C.byref(uid)), "desta_comm_create")
    return comm.value


def allreduce_grads(comm: int, grads: torch.Tensor) -> None:
    """In-place mean over the ranks of `comm` of a flat fp32 tensor, on the current stream (one RCCL all-reduce)."""
    assert grads.dtype == torch.float32 and grads.is_contiguous()
    lib.desta_allreduce_grads.argtypes = [vp, vp, i64, vp]
    check(lib.desta_allreduce_grads(comm, p(grads), grads.numel(), stream()), "desta_allreduce_grads")


def comm_destroy(comm: int) -> None:
    lib.desta_comm_destroy.argtypes = [vp]
    check(lib.desta_comm_destroy(comm), "desta_comm_destroy")


class Context:
    """`desta_create` / `desta_destroy` (include/desta_hip.h): optional per-device context.  Creating it checks the device is a
    gfx950 part, makes it current and creates the library's internal fork stream / events up front; closing it releases them."""

    def __init__(self, device: int = 0):
        lib.desta_create.argtypes = [i32, C.POINTER(vp)]
        h = vp()
        check(lib.desta_create(int(device), C.byref(h)), "desta_create")
        self.handle = h.value

    def info(self) -> dict:
        lib.desta_handle_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.c_char_p, C.c_size_t]
        dev, cus, arch = i32(), i32(), C.create_string_buffer(64)
        check(lib.desta_handle_info(self.handle, C.byref(dev), C.byref(cus), arch, 64), "desta_handle_info")
        return {"device": dev.value, "compute_units": cus.value, "arch": arch.value.decode()}

    def last_error(self) -> str:
        lib.desta_handle_last_error.argtypes = [vp]
        lib.desta_handle_last_error.restype = C.c_char_p
        return lib.desta_handle_last_error(self.handle).decode()

    def close(self) -> None:
        if self.handle:
            lib.desta_destroy.argtypes = [vp]
            check(lib.desta_destroy(self.handle), "desta_destroy")
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def attention_set_option(which: int, value: int) -> None:
    lib.desta_attention_set_option.argtypes = [i32, i32]
    check(lib.desta_attention_set_option(int(which), int(value)), "desta_attention_set_option")


def attention_set_concurrent_bwd(on: bool) -> None:
    lib.desta_attention_set_concurrent_bwd.argtypes = [i32]
    check(lib.desta_attention_set_concurrent_bwd(int(on)), "desta_attention_set_concurrent_bwd")


def _check_struct_layouts() -> None:
    lib.desta_sizeof_desc.restype = C.c_size_t
    lib.desta_sizeof_desc.argtypes = [i32]
    for which, cls in ((0, GemmDesc), (1, AttnDesc), (2, OptPlan)):
        if lib.desta_sizeof_desc(which) != C.sizeof(cls):
            raise ImportError(f"struct layout mismatch for {cls.__name__}: library {lib.desta_sizeof_desc(which)} B, binding {C.sizeof(cls)} B")


_check_struct_layouts()


_scatter = _sig("desta_scatter_rows_bf16", vp, vp, i32, i32, vp, vp)
_target_rows = _sig("desta_target_rows", vp, i32, i32, vp, vp, vp, i32, vp)


def scatter_rows(src, idx, rows, hidden, out):
    check(_scatter(p(src), p(idx), rows, hidden, p(out), stream()), "desta_scatter_rows_bf16")


def target_rows(labels, batch, seq, idx, compact_labels, count, s_major=False):
    """Rows with a real shifted target -> idx / compact label layout / device count int32[2] = (n, first position with a target)
    (see include/desta_hip.h)."""
    check(_target_rows(p(labels), batch, seq, p(idx), p(compact_labels), p(count), int(s_major), stream()), "desta_target_rows")


# ----------------------------------------------------------------------------- ORCA hybrid (ABI 6)
_orca_local_mix = _sig("desta_orca_local_mix", vp, vp, i32, i64, i32, vp, vp)
_orca_rope = _sig("desta_orca_rope", vp, vp, i32, i32, i32, C.c_float, C.c_float, i32, vp)
_orca_gate_residual = _sig("desta_orca_gate_residual", vp, i64, vp, vp, vp, vp, i64, i32, i32, vp, vp)
_orca_sim_loss = _sig("desta_orca_sim_loss", vp, vp, vp, i32, i32, i32, i64, i32, i32, vp, vp)
_orca_align = _sig("desta_orca_align", vp, i32, vp, i64, i64, i32, vp, i32, vp, vp)


def orca_local_mix(x, layer_weights, taps, rows, d, out):
    check(_orca_local_mix(p(x), p(layer_weights), taps, rows, d, p(out), stream()), "desta_orca_local_mix")


def orca_rope(x, y, batch, tokens, hidden, theta, position_scale, round_cos_sin=True):
    check(_orca_rope(p(x), p(y), batch, tokens, hidden, float(theta), float(position_scale), int(round_cos_sin), stream()), "desta_orca_rope")


def orca_gate_residual(hidden, ld_hidden, cross, gate_hidden, gate_w2, gate_b2, rows, hidden_size, gate_width, gate_out=None):
    check(_orca_gate_residual(p(hidden), ld_hidden, p(cross), p(gate_hidden), p(gate_w2), p(gate_b2), rows, hidden_size, gate_width,
                              p(gate_out), stream()), "desta_orca_gate_residual")


def orca_sim_loss(x, y, y_index, batch, nx, ny, y_rows, hidden, subtract_identity, partials):
    check(_orca_sim_loss(p(x), p(y), p(y_index), batch, nx, ny, y_rows, hidden, int(subtract_identity), p(partials), stream()), "desta_orca_sim_loss")


def orca_align(audio, tokens, hidden, row_stride, batch_stride, hidden_size, spans, n_spans, out):
    check(_orca_align(p(audio), tokens, p(hidden), row_stride, batch_stride, hidden_size, p(spans), n_spans, p(out), stream()), "desta_orca_align")


_orca_gate_residual_bwd = _sig("desta_orca_gate_residual_bwd", vp, i64, vp, vp, i64, i32, vp, vp, vp)
_orca_gate_mlp_bwd = _sig("desta_orca_gate_mlp_bwd", vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp)
_orca_align_bwd = _sig("desta_orca_align_bwd", vp, i32, vp, i64, i64, i32, vp, i32, C.c_float, vp, i64, i64, vp)
_orca_rope_bwd = _sig("desta_orca_rope_bwd", vp, i32, i32, i32, C.c_float, C.c_float, i32, i32, vp, vp, vp)
_orca_col2im_add = _sig("desta_orca_col2im_add", vp, i32, i32, i32, i32, i32, i32, vp, vp)
_orca_local_mix_bwd = _sig("desta_orca_local_mix_bwd", vp, vp, vp, i32, i64, i32, vp, vp, vp)
_orca_sim_loss_bwd = _sig("desta_orca_sim_loss_bwd", vp, vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, C.c_float, vp, vp)


def orca_gate_residual_bwd(d_out, ld, cross, gate, rows, hidden_size, d_cross, d_gate_pre):
    check(_orca_gate_residual_bwd(p(d_out), ld, p(cross), p(gate), rows, hidden_size, p(d_cross), p(d_gate_pre), stream()), "desta_orca_gate_residual_bwd")


_gate_ws: dict = {}


def orca_gate_mlp_bwd(d_gate_pre, gate_preact, gate_hidden, gate_w2, rows, gate_width, d_preact, d_w2, d_b2):
    key = (d_gate_pre.device, gate_width)
    if key not in _gate_ws:                                                   # 64 row slices x (gate_width + 1) partial sums
        _gate_ws[key] = torch.empty(64 * (gate_width + 1), dtype=torch.float32, device=d_gate_pre.device)
    check(_orca_gate_mlp_bwd(p(d_gate_pre), p(gate_preact), p(gate_hidden), p(gate_w2), rows, gate_width, p(d_preact), p(d_w2), p(d_b2), p(_gate_ws[key]),
                             stream()), "desta_orca_gate_mlp_bwd")


def orca_align_bwd(audio, tokens, hidden, row_stride, batch_stride, hidden_size, spans, n_spans, coef, d_hidden, d_row_stride, d_batch_stride):
    check(_orca_align_bwd(p(audio), tokens, p(hidden), row_stride, batch_stride, hidden_size, p(spans), n_spans, float(coef), p(d_hidden), d_row_stride,
                          d_batch_stride, stream()), "desta_orca_align_bwd")


def orca_rope_bwd(d_rotated, batch, tokens, hidden, theta, position_scale, round_cos_sin, n_first, d_first, d_rest):
    check(_orca_rope_bwd(p(d_rotated), batch, tokens, hidden, float(theta), float(position_scale), int(round_cos_sin), n_first, p(d_first), p(d_rest), stream()),
          "desta_orca_rope_bwd")


def orca_col2im_add(d_col, batch, tokens_out, tokens_padded, hidden, kernel, stride, d_padded):
    check(_orca_col2im_add(p(d_col), batch, tokens_out, tokens_padded, hidden, kernel, stride, p(d_padded), stream()), "desta_orca_col2im_add")


def orca_local_mix_bwd(d_out, x, layer_weights, taps, rows, d, d_layer_weights):
    ws = scratch(256 * 32, d_out.device, tag="orca_mix")
    check(_orca_local_mix_bwd(p(d_out), p(x), p(layer_weights), taps, rows, d, p(d_layer_weights), p(ws), stream()), "desta_orca_local_mix_bwd")


def orca_sim_loss_bwd(x, x_index, x_rows, y, y_index, y_rows, batch, nx, ny, hidden, subtract_identity, coef, d_x):
    check(_orca_sim_loss_bwd(p(x), p(x_index), x_rows, p(y), p(y_index), y_rows, batch, nx, ny, hidden, int(subtract_identity), float(coef), p(d_x), stream()),
          "desta_orca_sim_loss_bwd")
