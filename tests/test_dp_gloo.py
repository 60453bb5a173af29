"""CPU test of the N>1 data-parallel path: world_size 2 over gloo.

Each rank takes its shard of a global batch, computes the (oracle) gradient of its shard-mean loss,
sums the flat gradient through ``dp.allreduce_sum_`` and applies ``dp.grad_scale`` -- exactly what
``WaveNetEngine.allreduce_grads`` / ``optimizer_step`` do on GPUs -- and must land on the full-batch
gradient and on identical parameters after a TF-Adam step."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import wavenet_np as O
from tests._pkg import sub


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _flat(sp_like):
    return np.concatenate([a.reshape(-1) for _, a in O.flatten_named(sp_like, False)])


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dp = sub("dp")
    r, l, w = dp.init_from_env("gloo")
    assert (r, w) == (rank, world) and dp.world_size() == world
    dil = [1, 2, 4]
    GB, T, R, S, C = 4, 48, 8, 8, 16
    sp = O.init_stack_params(0, dil, 2, R, S, C, bias_scale=0.1)
    rng = np.random.default_rng(0)
    audio = rng.uniform(-1, 1, (GB, T)); codes = rng.integers(0, C, (GB, T))
    sl = dp.shard_batch(GB, rank, world)
    lg, cache = O.stack_forward(sp, audio[sl], shift_input=True)
    g, _ = O.stack_backward(sp, cache, O.dlogits_per_timestep(lg, codes[sl]))
    flat = torch.tensor(_flat(g))
    dp.allreduce_sum_(flat)
    flat *= dp.grad_scale()
    lg_full, cache_full = O.stack_forward(sp, audio, shift_input=True)
    g_full, _ = O.stack_backward(sp, cache_full, O.dlogits_per_timestep(lg_full, codes))
    assert np.allclose(flat.numpy(), _flat(g_full), rtol=1e-10, atol=1e-13)
    th, _, _ = O.adam_step_tf(_flat(sp), flat.numpy(), 0, 0, 1)
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, torch.tensor(th))
    assert torch.equal(gathered[0], gathered[1])      # replicas stay identical
    try:
        dp.shard_batch(5, rank, world)
        raise SystemExit("shard_batch accepted an indivisible batch")
    except ValueError:
        pass
    dist.barrier()
    dist.destroy_process_group()
    out.put(rank)


def test_dp_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def test_single_process_is_a_noop():
    dp = sub("dp")
    t = torch.ones(4)
    assert dp.allreduce_sum_(t) is None and dp.grad_scale() == 1.0 and dp.world_size() == 1
    assert dp.shard_batch(8, 1, 4) == slice(2, 4)
