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
LIB_PATH = os.path.join(_HERE, "lib", "libdesta_hip.so")

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
                ("preact", vp), ("ldp", i64), ("stride_p", i64), ("alpha", f32)]


lib.desta_abi_version.restype = i32
lib.desta_last_error.restype = C.c_char_p


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
         stride_r=0, stride_p=0):
    """out[M,N] = act(alpha * A[M,K] @ B[N,K]^T + bias) + residual  (bf16 operands, MFMA)."""
    d = GemmDesc()
    d.A, d.B, d.C = p(A), p(B), p(out)
    d.M, d.N, d.K, d.batch = M, N, K, batch
    d.lda = K if lda is None else lda
    d.ldb = K if ldb is None else ldb
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
    check(_gemm(C.byref(d), stream()), "desta_gemm_bf16_nt")
    return out


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
                ("sum_rows", i64), ("sum_cols", i64), ("max_batch", i32), ("max_cols", i32)]


lib.desta_adafactor_workspace_floats.restype = C.c_size_t
lib.desta_adafactor_workspace_floats.argtypes = [i32, i32, i64, i64, i64]
_adafactor = _sig("desta_clip_adafactor_step", C.POINTER(OptPlan), vp, vp, vp, vp, f32, f32, f32, f32, f32, vp)


def clip_adafactor_step(plan: OptPlan, params, grads, state, workspace, lr, beta2t, eps1, clip_threshold,
                        max_grad_norm):
    check(_adafactor(C.byref(plan), p(params), p(grads), p(state), p(workspace), lr, beta2t, eps1,
                     clip_threshold, max_grad_norm, stream()), "desta_clip_adafactor_step")
