"""bench.py launch contract: `python bench.py --gpus N` with N > 1 outside torchrun either launches N ranks
itself (fresh `torch.distributed.run` child, before any GPU call) or exits non-zero — it never prints a line for fewer ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_self_launch_starts_n_ranks_on_cpu():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check", "--backend", "gloo", "--single-device"],
                       env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _json_line(p.stdout)
    assert line["n_gpus"] == 2 and line["ranks_sum"] == 3.0 and line["backend"] == "gloo"
    assert "launching 2 ranks" in p.stderr


def test_more_ranks_than_gpus_exits_nonzero_without_a_line():
    import torch
    n = torch.cuda.device_count() + 1
    if n < 2:
        n = 2
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(max(n, 9))], env=ENV, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "refusing" in p.stderr


def test_world_size_mismatch_exits_nonzero():
    env = dict(ENV, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launch-check"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 2 and "does not match WORLD_SIZE" in p.stderr and not p.stdout.strip()


@pytest.mark.gpu
def test_two_rank_bench_on_one_card_reports_the_exchange():
    """The real bench through its own launcher at N = 2 (two ranks share cuda:0 over gloo: the box has one GPU): n_gpus == 2,
    the all-reduce / wait_update fields are present, value = steps / time (optimizer steps of the node), gpu_batch_steps_per_s = 2 x that."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-pass", "--no-calibration"],
                       env=dict(ENV, HSA_ENABLE_IPC_MODE_LEGACY="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _json_line(p.stdout)
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 16
    assert line["bytes_allreduced"] > 5e8 and line["allreduce_ms"] > 0 and line["wait_update_stall_ms"] >= 0
    assert line["comm_backend"] == "gloo"
    # value = OPTIMIZER steps per second of the node (global batch 8 x N per step); the per-GPU aggregates sit beside it
    per_s = 1.0 / (line["ms_per_step"] * 1e-3)
    assert abs(line["value"] - per_s) < 1e-6 and abs(line["gpu_batch_steps_per_s"] - 2 * per_s) < 1e-6
    assert abs(line["samples_per_s"] - 16 * per_s) < 1e-5
