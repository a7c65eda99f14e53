"""N>1 path on CPU: world_size-2 `gloo` processes exercise the data-parallel exchange of the flat
gradient arena (mean over ranks of per-rank token-mean gradients, hazard H8) and check it against a
single process that sees both micro-batches.  The HIP optimizer itself is covered by the -m gpu tests;
here the arena lives on CPU and the oracle supplies the per-rank gradients."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, empty_rank=-1):
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "desta2.5-audio_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import desta_oracle as O
    from helpers import cfg_from_dims
    from desta.models.modeling_desta25 import connector_param_shapes
    from desta.optim import ParamArena
    from desta.trainer.desta_trainer import allreduce_mean_
    torch.set_num_threads(2)
    d = O.tiny_dims(False)
    w = O.init_weights(d, seed=7)
    arena = ParamArena(list(connector_param_shapes(cfg_from_dims(d)).items()), "cpu")
    batch = O.synthetic_batch(d, B=1, S_ctx=4, S_tgt=8 + 4 * rank, seed=50 + rank)     # ragged target lengths per rank
    names = O.trainable_names(d)
    for n in names:
        w[n].requires_grad_(True)
    if rank == empty_rank:
        # `_empty_batch` on this rank only (audio decode errors): it still enters the collective with a zero arena, as
        # DeSTA25Trainer.training_step does — skipping it would block the other rank in all_reduce forever
        loss = torch.zeros(())
    else:
        loss, _ = O.model_forward(w, d, batch)
        loss.backward()
        for n in names:
            arena.grad(n).copy_(w[n].grad)
    allreduce_mean_(arena.grads)
    q.put((rank, float(loss.detach()), arena.grads.clone().numpy()))       # by value: a torch tensor travels as a shared-memory handle the parent must fetch while this process is alive
    dist.barrier()
    dist.destroy_process_group()
    q.close()
    q.join_thread()
    os._exit(0)                                                              # skip interpreter teardown: the result is out


@pytest.mark.parametrize("empty_rank", [-1, 1])
def test_two_rank_gradient_mean_matches_single_process(empty_rank):
    import socket
    with socket.socket() as sk:                          # a port nobody holds right now (a fixed formula can hit one in TIME_WAIT)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, empty_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    res = [(r, l, torch.from_numpy(g)) for r, l, g in res]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, p.exitcode
    assert torch.equal(res[0][2], res[1][2]), "ranks disagree after the all-reduce"
    # single process: mean over ranks of the per-rank gradients
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "desta2.5-audio_amd")):
        sys.path.insert(0, p)
    import desta_oracle as O
    from helpers import cfg_from_dims
    from desta.models.modeling_desta25 import connector_param_shapes
    from desta.optim import ParamArena
    d = O.tiny_dims(False)
    ref = ParamArena(list(connector_param_shapes(cfg_from_dims(d)).items()), "cpu")
    # fp32 CPU kernels split their reductions by thread count and load: the single-process rerun is not bit-identical to the
    # workers' (2 threads each), so the comparison carries a few fp32 ulps of slack (the ranks themselves must agree exactly)
    for rank in range(world):
        w = O.init_weights(d, seed=7)
        names = O.trainable_names(d)
        for n in names:
            w[n].requires_grad_(True)
        batch = O.synthetic_batch(d, B=1, S_ctx=4, S_tgt=8 + 4 * rank, seed=50 + rank)
        if rank == empty_rank:
            assert res[rank][1] == 0.0
            continue
        loss, _ = O.model_forward(w, d, batch)
        assert abs(float(loss.detach()) - res[rank][1]) < 2e-5
        loss.backward()
        for n in names:
            ref.grad(n).add_(w[n].grad / world)
    # ... measured against the arena's largest gradient (an element-wise atol trips over near-zero entries whose fp32 sums were
    # split differently); a wrong reduction (SUM instead of mean, a missing rank) is off by a factor, not by 1e-4
    err = float((res[0][2] - ref.grads).abs().max()) / float(ref.grads.abs().max())
    assert err < 1e-4, err
