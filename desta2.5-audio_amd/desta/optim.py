"""Flat fp32 parameter arena + fused clip/Adafactor (host side of csrc/adafactor.hip).

Mirrors what the reference gets from HF Trainer: `clip_grad_norm_(1.0)` then
`transformers.optimization.Adafactor(lr, scale_parameter=False, relative_step=False)` with the
two weight-decay groups (TF:trainer.py:1181-1195, 1305-1315, 1780-1797; TF:trainer_optimizer.py:197),
and `get_linear_schedule_with_warmup` (TF:optimization.py:101-104, train_desta.py:143).
"""
from __future__ import annotations

import ctypes
import math
from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import torch

from . import _hip

ALIGN = 64           # floats; keeps every tensor 256-B aligned in the arena
UNIT_ROWS = 64
CHUNK = 16384        # elements per work item of the update kernels (256 threads x 16 float4, csrc/adafactor.hip)
GROUP_FLOATS = 16 << 20   # <= 64 MB of gradients per (sum u^2, apply) launch pair: g + p read + p write of a group = 192 MB < 256 MB Infinity Cache


def _al(n: int, a: int = ALIGN) -> int:
    return (n + a - 1) // a * a


class ParamArena:
    """All trainable tensors (and their gradients) live in two contiguous fp32 buffers, so the
    optimizer and the data-parallel all-reduce see ONE array (one RCCL call, no bucketing)."""

    def __init__(self, named_shapes: Sequence[Tuple[str, Sequence[int]]], device):
        self.names: List[str] = []
        self.shapes: Dict[str, Tuple[int, ...]] = {}
        self.offsets: Dict[str, int] = {}
        off = 0
        for name, shape in named_shapes:
            self.names.append(name)
            self.shapes[name] = tuple(int(s) for s in shape)
            self.offsets[name] = off
            off += _al(int(math.prod(shape)))
        self.numel = off
        self.params = torch.zeros(off, dtype=torch.float32, device=device)
        self.grads = torch.zeros(off, dtype=torch.float32, device=device)

    def _view(self, buf, name):
        o = self.offsets[name]
        n = int(math.prod(self.shapes[name]))
        return buf[o:o + n].view(self.shapes[name])

    def param(self, name: str) -> torch.Tensor:
        return self._view(self.params, name)

    def grad(self, name: str) -> torch.Tensor:
        return self._view(self.grads, name)

    def true_numel(self) -> int:
        return sum(int(math.prod(s)) for s in self.shapes.values())


def decay_mask(names: Sequence[str]) -> List[bool]:
    """HF Trainer decay group (TF:trainer.py:1305-1315): not inside an nn.LayerNorm module and no
    'bias' / norm pattern in the name.  For the connector: LayerNorm modules are `*.LayerNorm` and
    `proj.0`; `layer_prompts` and `layer_weights` ARE decayed."""
    out = []
    for n in names:
        nd = (("bias" in n) or ("LayerNorm" in n) or (".proj.0." in n) or ("layernorm" in n.lower()) or ("_norm" in n)
              or (".global_proj.0." in n) or (".local_ln." in n) or n.endswith(".ln.weight"))          # ORCA: nn.LayerNorm modules under other names
        out.append(not nd)
    return out


def linear_warmup_lr(step: int, base_lr: float, warmup: int, total: int) -> float:
    """LR after `step` scheduler steps (get_linear_schedule_with_warmup)."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    return base_lr * max(0.0, (total - step) / max(1, total - warmup))


class FusedAdafactor:
    def __init__(self, arena: ParamArena, weight_decay: float = 0.01, eps1: float = 1e-30,
                 clip_threshold: float = 1.0, decay_rate: float = -0.8, max_grad_norm: float = 1.0,
                 decay: Sequence[bool] = None):
        self.arena = arena
        self.eps1, self.clip_threshold, self.decay_rate = eps1, clip_threshold, decay_rate
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        dev = arena.params.device
        decay = decay_mask(arena.names) if decay is None else list(decay)
        tensors, twd, units, ucol, vecs, vwd = [], [], [], [], [], []
        self.state_slices: "OrderedDict[str, dict]" = OrderedDict()
        st_off, col_ws_off, sum_rows, sum_cols, max_batch, max_cols = 0, 0, 0, 0, 1, 1
        for name, dk in zip(arena.names, decay):
            shape = arena.shapes[name]
            wd = weight_decay if dk else 0.0
            if len(shape) >= 2:
                R, Cn = shape[-2], shape[-1]
                nb = int(math.prod(shape[:-2]))
                row_off = st_off
                st_off += _al(nb * R, 4)
                col_off = st_off
                st_off += _al(nb * Cn, 4)
                # rows per statistics unit: 64; a tensor with ragged rows (cols % 4 != 0, in practice a Conv1d weight [out, in, k]: `in` rows of
                # k columns per batch item) takes up to 32 K elements per unit instead — 64 x 5 elements per block would mean 262 144 blocks
                # for ORCA's [4096, 4096, 5] local_conv.weight
                ur = UNIT_ROWS if Cn % 4 == 0 else max(UNIT_ROWS, min(R, 32768 // Cn))
                upb = (R + ur - 1) // ur
                unit0 = len(units)
                for b in range(nb):
                    for k in range(upb):
                        r0 = k * ur
                        units.append([len(tensors), b, r0, min(ur, R - r0)])
                        ucol.append(col_ws_off)
                        col_ws_off += _al(Cn, 4)
                tensors.append([arena.offsets[name], nb, R, Cn, row_off, col_off, unit0, nb * upb])
                twd.append(wd)
                self.state_slices[name] = {"row": (row_off, shape[:-1]), "col": (col_off, shape[:-2] + shape[-1:])}
                sum_rows = max(sum_rows, row_off + nb * R)
                sum_cols = max(sum_cols, col_off + nb * Cn)
                max_batch, max_cols = max(max_batch, nb), max(max_cols, Cn)
            else:
                n = int(math.prod(shape))
                vecs.append([arena.offsets[name], n, st_off])
                vwd.append(wd)
                self.state_slices[name] = {"sq": (st_off, shape)}
                st_off += _al(n, 4)
        # the kernels index rowsum/rfac/cfac workspaces with the STATE offsets, so size them by st_off
        self.state = torch.zeros(max(st_off, 4), dtype=torch.float32, device=dev)
        i64, i32, f32 = torch.int64, torch.int32, torch.float32
        self._tensors = torch.tensor(tensors or [[0] * 8], dtype=i64, device=dev)
        self._twd = torch.tensor(twd or [0.0], dtype=f32, device=dev)
        self._units = torch.tensor(units or [[0] * 4], dtype=i32, device=dev)
        self._ucol = torch.tensor(ucol or [0], dtype=i64, device=dev)
        self._vecs = torch.tensor(vecs or [[0] * 3], dtype=i64, device=dev)
        self._vwd = torch.tensor(vwd or [0.0], dtype=f32, device=dev)
        pl = _hip.OptPlan()
        pl.tensors, pl.tensor_wd, pl.n_tensors = self._tensors.data_ptr(), self._twd.data_ptr(), len(tensors)
        pl.units, pl.unit_col_off, pl.n_units = self._units.data_ptr(), self._ucol.data_ptr(), len(units)
        pl.vecs, pl.vec_wd, pl.n_vec = self._vecs.data_ptr(), self._vwd.data_ptr(), len(vecs)
        pl.sum_rows, pl.sum_cols = st_off, st_off
        pl.max_batch, pl.max_cols = max_batch, max_cols
        # work items of the update kernels: <= CHUNK contiguous elements of one [rows, cols] matrix; tensors in DESCENDING
        # arena order (the update pass starts where the statistics pass ended: that tail is still in the Infinity Cache),
        # the chunks of one tensor contiguous
        # Tensors with ragged rows (cols % 4 != 0: the ORCA Conv1d weight [h, h, 5]) get no chunks: the unit-based kernels update them
        # over their own unit range (`ragged_units`, ABI 7)
        chunks, ten_chunks, fin = [], [[0, 0] for _ in tensors], []
        ragged = [ti for ti, t in enumerate(tensors) if t[3] % 4 != 0]
        for ti in reversed(range(len(tensors))):
            _, nb, R, Cn = tensors[ti][:4]
            if ti in ragged:
                continue
            ten_chunks[ti][0] = len(chunks)
            for b in range(nb):
                for e0 in range(0, R * Cn, CHUNK):
                    chunks.append([ti, b, e0, min(CHUNK, R * Cn - e0)])
            ten_chunks[ti][1] = len(chunks) - ten_chunks[ti][0]
        for ti, (_, nb, R, Cn, *_rest) in enumerate(tensors):
            for b in range(nb):
                fin += [[ti, b, part] for part in range(1 + (Cn + 255) // 256)]
        self._chunks = torch.tensor(chunks or [[0] * 4], dtype=i32, device=dev)
        self._ten_chunks = torch.tensor(ten_chunks or [[0, 0]], dtype=i32, device=dev)
        self._fin = torch.tensor(fin or [[0] * 3], dtype=i32, device=dev)
        pl.chunks, pl.ten_chunks, pl.n_chunks = self._chunks.data_ptr(), self._ten_chunks.data_ptr(), len(chunks)
        pl.max_chunks_per_tensor = max([c[1] for c in ten_chunks] or [0])
        pl.fin, pl.n_fin, pl.colpart_floats = self._fin.data_ptr(), len(fin), col_ws_off
        pl.cols_multiple_of_4 = 1
        rag = [v for ti in ragged for v in (tensors[ti][6], tensors[ti][7])]
        self._ragged = (ctypes.c_int32 * max(len(rag), 1))(*rag)                # HOST array, kept alive with the plan
        pl.ragged_units, pl.n_ragged = ctypes.cast(self._ragged, ctypes.c_void_p), len(ragged)
        # launch groups: cut the chunk list at tensor boundaries (a tensor's rms needs all of its chunk sums before its apply)
        bounds, acc = [0], 0
        for ti in reversed(range(len(tensors))):
            if ti in ragged:
                continue
            n = tensors[ti][1] * tensors[ti][2] * tensors[ti][3]
            if acc and acc + n > GROUP_FLOATS:
                bounds.append(ten_chunks[ti][0])
                acc = 0
            acc += n
        bounds.append(len(chunks))
        self._group_bounds = (ctypes.c_int32 * len(bounds))(*bounds)            # HOST array, kept alive with the plan
        pl.group_bounds, pl.n_groups = ctypes.cast(self._group_bounds, ctypes.c_void_p), len(bounds) - 1
        self.plan = pl
        nws = _hip.lib.desta_adafactor_workspace_floats_v3(ctypes.byref(pl), col_ws_off)
        self.workspace = torch.zeros(nws, dtype=torch.float32, device=dev)

    def step(self, lr: float) -> None:
        """clip_grad_norm_(max_grad_norm) + Adafactor update, in place on the arena."""
        self.step_count += 1
        beta2t = 1.0 - math.pow(self.step_count, self.decay_rate)
        _hip.clip_adafactor_step(self.plan, self.arena.params, self.arena.grads, self.state, self.workspace,
                                 lr, beta2t, self.eps1, self.clip_threshold, self.max_grad_norm)

    def grad_norm(self) -> torch.Tensor:
        """Pre-clip global gradient norm of the last step (device scalar, no sync)."""
        return self.workspace[0]

    # -- wire format of `transformers.optimization.Adafactor.state_dict()` as HF Trainer writes it to
    #    checkpoint-<step>/optimizer.pt: two param groups (decay first, TF:trainer.py:1181-1195), parameters
    #    numbered in the reference's named_parameters() order inside each group
    def hf_state_dict(self, ref_names: Sequence[str], lr: float, weight_decay: float = 0.01) -> dict:
        dm = dict(zip(ref_names, decay_mask(ref_names)))
        order = [n for n in ref_names if dm[n]] + [n for n in ref_names if not dm[n]]
        n_decay = sum(dm.values())
        state = {}
        for idx, name in enumerate(order):
            sl, p = self.state_slices[name], self.arena.param(name)
            ent = {"step": self.step_count, "RMS": (p.norm(2) / math.sqrt(p.numel())).cpu()}
            for key, hf in (("row", "exp_avg_sq_row"), ("col", "exp_avg_sq_col"), ("sq", "exp_avg_sq")):
                if key in sl:
                    off, shape = sl[key]
                    ent[hf] = self.state[off:off + int(math.prod(shape))].view(tuple(shape)).detach().cpu().clone()
            state[idx] = ent
        common = {"lr": lr, "eps": (self.eps1, 1e-3), "clip_threshold": self.clip_threshold, "decay_rate": self.decay_rate,
                  "beta1": None, "scale_parameter": False, "relative_step": False, "warmup_init": False}
        groups = [dict(common, weight_decay=weight_decay, params=list(range(n_decay))),
                  dict(common, weight_decay=0.0, params=list(range(n_decay, len(order))))]
        return {"state": state if self.step_count > 0 else {}, "param_groups": groups}

    def load_hf_state_dict(self, sd: dict, ref_names: Sequence[str]) -> None:
        dm = dict(zip(ref_names, decay_mask(ref_names)))
        order = [n for n in ref_names if dm[n]] + [n for n in ref_names if not dm[n]]
        self.step_count = 0
        for idx, name in enumerate(order):
            ent = sd["state"].get(idx)
            if ent is None:
                continue
            self.step_count = int(ent["step"])
            for key, hf in (("row", "exp_avg_sq_row"), ("col", "exp_avg_sq_col"), ("sq", "exp_avg_sq")):
                if key in self.state_slices[name]:
                    off, shape = self.state_slices[name][key]
                    self.state[off:off + int(math.prod(shape))].copy_(ent[hf].reshape(-1))

    # -- HF `optimizer.pt`-shaped state (exp_avg_sq_row / exp_avg_sq_col / exp_avg_sq, step)
    def state_dict(self) -> dict:
        st = {}
        for i, name in enumerate(self.arena.names):
            sl, ent = self.state_slices[name], {"step": self.step_count, "RMS": 0}
            for key, hf in (("row", "exp_avg_sq_row"), ("col", "exp_avg_sq_col"), ("sq", "exp_avg_sq")):
                if key in sl:
                    off, shape = sl[key]
                    ent[hf] = self.state[off:off + int(math.prod(shape))].view(tuple(shape)).clone()
            st[i] = ent
        return {"state": st, "names": list(self.arena.names), "step": self.step_count}

    def load_state_dict(self, sd: dict) -> None:
        self.step_count = int(sd.get("step", 0))
        for i, name in enumerate(self.arena.names):
            ent = sd["state"][i]
            sl = self.state_slices[name]
            for key, hf in (("row", "exp_avg_sq_row"), ("col", "exp_avg_sq_col"), ("sq", "exp_avg_sq")):
                if key in sl:
                    off, shape = sl[key]
                    self.state[off:off + int(math.prod(shape))].copy_(ent[hf].reshape(-1))
