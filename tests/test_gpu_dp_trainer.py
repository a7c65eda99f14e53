"""GPU: the data-parallel TRAINER path end to end on one card: two fresh child processes share cuda:0 and
talk over `gloo` (the driver's box has one GPU; RCCL needs one device per rank).  Each child runs `DeSTA25Trainer.training_step`
on its own batches with `overlap_comm=True` — side-stream flat all-reduce, fused clip + Adafactor, weight re-cast, next-batch
Whisper prefetch — with Q-Former dropout ON and, at one step, an `_empty_batch` on rank 1 only (the reference deadlocks there
under DDP; here the rank contributes zeros and stays in the collective).  Asserted: both ranks end with BIT-IDENTICAL parameters
and optimizer state, equal bit for bit to a single process that averages the two ranks' gradient arenas itself before one
optimizer step (accelerate DDP mean-of-means semantics, hazard H8)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, EMPTY_AT = 4, (2, 1)                      # rank 1 draws an empty batch at step index 2
LR, WARM, TOTAL = 1e-3, 2, 10


def _setup_paths():
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "desta2.5-audio_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _batches(d, rank):
    import desta_oracle as O
    out = []
    for i in range(STEPS):
        if (i, rank) == EMPTY_AT:
            out.append({"_empty_batch": True})
        else:
            out.append(O.synthetic_batch(d, B=2, S_ctx=4 + rank, S_tgt=10 + 3 * i, seed=100 + 10 * i + rank, pad=[rank, 0]))
    return out


def _eval_batches(d):
    import desta_oracle as O
    return [O.synthetic_batch(d, B=2, S_ctx=4, S_tgt=9 + i, seed=900 + i) for i in range(5)]


def _model():
    import desta_oracle as O
    from helpers import cfg_from_dims
    from desta.models.modeling_desta25 import DeSTA25AudioModel
    d = O.tiny_dims(False)
    return d, DeSTA25AudioModel(cfg_from_dims(d, dropout=0.1), weights=O.init_weights(d, seed=7))


def worker(out_dir):
    _setup_paths()
    import torch.distributed as dist
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    if os.environ.get("DESTA_TEST_BACKEND") == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments
    d, model = _model()
    tr = DeSTA25Trainer(model, args=TrainingArguments(learning_rate=LR, warmup_steps=WARM, max_steps=TOTAL, logging_steps=1, overlap_comm=True))
    assert tr.world == world and tr._side is not None and model.dropout_seed == 1 + rank
    losses = tr.train(_batches(d, rank))
    assert tr.global_step == STEPS and tr.optimizer.step_count == STEPS
    if os.environ.get("DESTA_TEST_EVAL"):
        # data-parallel evaluate(): every "sample" of the eval set is one collated batch (collator = identity on a 1-row list);
        # rank r takes batches r, r + world, ...; loss sums are all-reduced, result files written by rank 0 only
        import types
        tr.eval_dataset, tr.data_collator = _eval_batches(d), (lambda rows: rows[0])
        tr.args.per_device_eval_batch_size = 1
        tr.cfg = types.SimpleNamespace(exp_dir=out_dir, get=lambda k, dflt=None: out_dir if k == "exp_dir" else dflt)
        metrics = tr.evaluate()
        torch.save(metrics, os.path.join(out_dir, f"eval{rank}.pt"))
    from desta.trainer.desta_trainer import ALLREDUCE_CALLS
    torch.save({"params": model.arena.params.cpu(), "state": tr.optimizer.state.cpu(), "losses": losses,
                "backend": dist.get_backend(), "allreduce_calls": dict(ALLREDUCE_CALLS)}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_trainer_bit_identical_and_equals_single_process_mean(tmp_path):
    import socket
    with socket.socket() as sk:                          # a port nobody holds right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(tmp_path)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt", weights_only=True) for r in range(2))
    assert r0["allreduce_calls"] == r1["allreduce_calls"] == {"gloo:SUM*1/world": STEPS}
    assert torch.equal(r0["params"], r1["params"]), "ranks diverged"
    assert torch.equal(r0["state"], r1["state"]), "optimizer state diverged across ranks"
    assert r1["losses"][EMPTY_AT[0]] == 0.0 and all(l > 0 for l in r0["losses"])

    # single process: per step, each rank's gradient arena from its own forward/backward, mean of the two, ONE optimizer step
    _setup_paths()
    from desta.optim import FusedAdafactor, linear_warmup_lr
    d, model = _model()
    opt = FusedAdafactor(model.arena, weight_decay=0.01, max_grad_norm=1.0)
    data = [_batches(d, r) for r in range(2)]
    fwd_count = [0, 0]
    model.train()
    for i in range(STEPS):
        grads = []
        for r in range(2):
            if data[r][i].get("_empty_batch"):
                grads.append(torch.zeros_like(model.arena.grads))
                continue
            model.dropout_seed, model._fwd_count = 1 + r, fwd_count[r]          # the dropout stream of that rank at that step
            out = model(**data[r][i])
            assert abs(float(out.loss) - (r0, r1)[r]["losses"][i]) == 0.0, (i, r)
            model.backward()
            fwd_count[r] += 1
            grads.append(model.arena.grads.clone())
        model.arena.grads.copy_(grads[0] + grads[1]).mul_(0.5)                     # gloo: SUM then * 1/world
        opt.step(linear_warmup_lr(i, LR, WARM, TOTAL))
        model.connector.refresh_weights()
    torch.cuda.synchronize()
    assert torch.equal(model.arena.params.cpu(), r0["params"]), float((model.arena.params.cpu() - r0["params"]).abs().max())
    assert torch.equal(opt.state.cpu(), r0["state"])


@pytest.mark.gpu
def test_two_rank_evaluate_shards_reduces_and_writes_once(tmp_path):
    """`DeSTA25Trainer.evaluate()` under data parallel: the eval batches are sharded by rank, the loss sums are
    all-reduced, rank 0 alone writes the result files: both ranks report the same eval_loss / eval_ppl, equal to one process
    evaluating every batch, and <exp_dir>/results/val holds exactly ONE report."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0", DESTA_TEST_EVAL="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(tmp_path)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    m0, m1 = (torch.load(tmp_path / f"eval{r}.pt", weights_only=True) for r in range(2))
    assert m0 == m1 and m0["eval_loss"] > 0
    reports = [f for f in os.listdir(tmp_path / "results" / "val") if f.endswith("-report.json")]
    assert len(reports) == 1, reports
    # single process over all five batches, same trained parameters
    _setup_paths()
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments
    d, model = _model()
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    model.arena.params.copy_(r0["params"].cuda())
    model.mark_weights_updated()
    tr = DeSTA25Trainer(model, args=TrainingArguments(overlap_comm=False))
    ref = tr.evaluate(eval_batches=_eval_batches(d))
    assert abs(ref["eval_loss"] - m0["eval_loss"]) < 1e-6 and abs(ref["eval_ppl"] - m0["eval_ppl"]) < 1e-4


@pytest.mark.gpu
def test_rccl_branch_at_world_size_one_is_the_identity(tmp_path):
    """The `nccl` (= RCCL) branch of `allreduce_mean_` — `all_reduce(AVG)` of the flat fp32 arena on the trainer's side stream —
    needs one device per rank, so on a one-GPU box it runs at world size 1 (forced through the collective): RCCL communicator
    with `device_id`, the AVG op on this ROCm build, stream ordering against the fused optimizer.  AVG over one rank is the
    identity: parameters / optimizer state equal an undistributed trainer bit for bit."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", DESTA_TEST_BACKEND="nccl",
               DESTA_ALLREDUCE_WORLD1="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(tmp_path)], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    assert r0["backend"] == "nccl" and r0["allreduce_calls"] == {"nccl:AVG": STEPS}, (r0["backend"], r0["allreduce_calls"])   # the RCCL AVG branch ran, once per step
    _setup_paths()
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments
    d, model = _model()
    tr = DeSTA25Trainer(model, args=TrainingArguments(learning_rate=LR, warmup_steps=WARM, max_steps=TOTAL, logging_steps=1, overlap_comm=True))
    losses = tr.train(_batches(d, 0))
    assert losses == r0["losses"]
    assert torch.equal(model.arena.params.cpu(), r0["params"]) and torch.equal(tr.optimizer.state.cpu(), r0["state"])


@pytest.mark.gpu
def test_gradient_accumulation_equals_the_sum_of_micro_batch_gradients():
    """`trainer.accumulate_grad_batches: 2` (HF gradient_accumulation_steps): two micro-batches per optimizer step, the update is
    made from the SUM of their gradient arenas — what the reference does: its `forward(**kwargs)` makes HF treat the model as
    accepting loss kwargs, `compute_loss` drops `num_items_in_batch`, so the loss is never divided by the window length
    (TF:trainer.py:1952-1954, SURVEY hazard H8) — bit for bit what one gets by adding the two arenas by hand; `global_step`
    counts optimizer steps; an empty micro-batch contributes nothing."""
    _setup_paths()
    from desta.optim import FusedAdafactor, linear_warmup_lr
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments
    d, model = _model()
    data = [b for b in _batches(d, 0)] + [{"_empty_batch": True}, _batches(d, 1)[0]]          # 4 real micro-batches + (empty, real)
    tr = DeSTA25Trainer(model, args=TrainingArguments(learning_rate=LR, warmup_steps=WARM, max_steps=TOTAL, logging_steps=1,
                                                      gradient_accumulation_steps=2, overlap_comm=False))
    losses = tr.train(data)
    assert len(losses) == 6 and tr.global_step == 3 and tr.optimizer.step_count == 3 and losses[4] == 0.0
    got_p, got_s = model.arena.params.clone(), tr.optimizer.state.clone()

    d, ref = _model()
    opt = FusedAdafactor(ref.arena, weight_decay=0.01, max_grad_norm=1.0)
    ref.train()
    ref.dropout_seed = 1                                                       # what the trainer sets on rank 0
    for stp in range(3):
        acc = torch.zeros_like(ref.arena.grads)
        for b in data[2 * stp:2 * stp + 2]:
            if b.get("_empty_batch"):
                continue
            ref(**b)
            ref.backward()
            acc += ref.arena.grads
        ref.arena.grads.copy_(acc)
        opt.step(linear_warmup_lr(stp, LR, WARM, TOTAL))
        ref.connector.refresh_weights()
    torch.cuda.synchronize()
    assert torch.equal(got_p, ref.arena.params) and torch.equal(got_s, opt.state)


@pytest.mark.gpu
def test_c_abi_allreduce_grads_at_world_size_one():
    """desta_comm_* / desta_allreduce_grads (the exchange for hosts without torch.distributed): RCCL communicator of ONE rank
    (the box has one GPU: RCCL wants a device per rank), ncclAvg of a 131.5 M-float arena on a side stream: the identity, bit
    for bit; a second communicator can be made after the first is destroyed; bad arguments are reported, not thrown."""
    _setup_paths()
    from desta import _hip as H
    uid = H.comm_get_unique_id()
    assert len(uid) == 128
    comm = H.comm_create(1, 0, uid)
    g = torch.randn(131_540_000 // 64 * 64, device="cuda")
    ref = g.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        H.allreduce_grads(comm, g)
        H.allreduce_grads(comm, g)
    side.synchronize()
    assert torch.equal(g, ref)
    H.comm_destroy(comm)
    comm2 = H.comm_create(1, 0, H.comm_get_unique_id())
    H.allreduce_grads(comm2, g[:1024])
    torch.cuda.synchronize()
    assert torch.equal(g, ref)
    H.comm_destroy(comm2)
    with pytest.raises(RuntimeError):
        H.comm_create(2, 5, uid)                                   # rank outside the world


if __name__ == "__main__" and len(sys.argv) >= 3 and sys.argv[1] == "--worker":
    worker(sys.argv[2])


@pytest.mark.gpu
def test_stream_overlaps_are_bit_identical_to_the_in_line_step():
    """Round 4: `overlap_encoder` (the frozen Whisper forward of batch t+1 on its own stream beside the connector / LLM of batch t)
    is the DEFAULT; `overlap_connector_backward` (the connector's backward of step t on the optimizer's side stream) is an option.  Six
    optimizer steps with Q-Former dropout on, ragged batches and an `_empty_batch` in the middle: the losses, the parameters and
    the optimizer state equal, bit for bit, the same steps run with no side stream at all.  (The frozen encoder depends on
    nothing the optimizer writes, /root/reference/desta/models/modeling_desta25.py:577-598.)"""
    _setup_paths()
    import desta_oracle as O
    from desta.trainer.desta_trainer import DeSTA25Trainer, TrainingArguments
    assert TrainingArguments().overlap_encoder and TrainingArguments().overlap_comm
    results = []
    for on in (True, False):
        d, model = _model()
        tr = DeSTA25Trainer(model, args=TrainingArguments(learning_rate=LR, warmup_steps=WARM, max_steps=TOTAL, logging_steps=1,
                                                          overlap_comm=on, overlap_encoder=on, overlap_connector_backward=on))
        assert (tr._side is not None) == on and (tr._enc_stream is not None) == on
        batches = [O.synthetic_batch(d, B=2 + (i % 2), S_ctx=4 + i, S_tgt=9 + 2 * i, seed=300 + i, pad=[i % 3, 0, 1][:2 + (i % 2)]) for i in range(6)]
        batches[3] = {"_empty_batch": True}
        losses = tr.train(batches)
        torch.cuda.synchronize()
        results.append((losses, model.arena.params.clone(), tr.optimizer.state.clone(), model.arena.grads.clone()))
    (l1, p1, s1, g1), (l0, p0, s0, g0) = results
    assert l1 == l0, (l1, l0)
    assert torch.equal(g1, g0), "gradient arena differs between the overlapped and the in-line step"
    assert torch.equal(p1, p0) and torch.equal(s1, s0)
