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


def _student_worker(rank, world, port, out):
    """Student under data parallelism: loss divides by the LOCAL batch (model.py:379), so the all-reduced
    SUM times 1/world is the global-batch gradient; tf.clip_by_global_norm is applied to THAT (model.py:384-385), and
    the squared-norm power loss is additive over clips, so no extra collective is needed.  Also the auto-encoder's
    SUM loss (ops.py:173-174): shard gradients simply add."""
    from oracle import wavenet_torch as OT
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dp = sub("dp")
    dp.init_from_env("gloo")
    GB, T, R, S, E, pool, M = 4, 768, 8, 16, 3, 64, 3
    dil = [1, 2, 4]
    flows = [O.init_flow_params(10 + i, dil, 2, R, S, E, bias_scale=0.1) for i in range(2)]
    for p in flows:
        p.head_w2 = p.head_w2 * 0.3
    rng = np.random.default_rng(0)
    noise = rng.logistic(0, 1, (GB, T)) * 0.15
    cond = rng.standard_normal((GB, T // pool, E))
    truth = O.synthetic_audio(GB, T, seed=1).astype(np.float64)
    tl = rng.standard_normal((GB, T, 4 * M)) * 0.5
    abg = (0.9, 1.1, 0.02)

    def grads(sl):
        ts = [OT.TorchStack(p) for p in flows]
        t = lambda a: torch.tensor(a[sl])
        OT.student_loss(ts, t(noise), t(cond), pool, t(tl), t(truth), *abg)["loss"].backward()
        return torch.cat([v.grad.reshape(-1) for st in ts for _, v in OT.flow_named(st)])

    flat = grads(dp.shard_batch(GB, rank, world))
    dp.allreduce_sum_(flat)
    flat *= dp.grad_scale()
    full = grads(slice(0, GB))
    assert torch.allclose(flat, full, rtol=1e-9, atol=1e-12)
    c_dp, n_dp = O.clip_by_global_norm([flat.numpy()], 1.0)
    c_full, n_full = O.clip_by_global_norm([full.numpy()], 1.0)
    assert abs(n_dp - n_full) < 1e-9 * n_full and np.allclose(c_dp[0], c_full[0], rtol=1e-9, atol=1e-12)
    # clipping each shard's gradient BEFORE the all-reduce would be a different update (the slow `train` path of the
    # reference, model.py:603-632, which student.py does not use)
    local = grads(dp.shard_batch(GB, rank, world))
    pre = torch.tensor(O.clip_by_global_norm([local.numpy()], 1.0)[0][0])
    dp.allreduce_sum_(pre)
    assert not np.allclose((pre * dp.grad_scale()).numpy(), c_full[0], rtol=1e-3)

    # auto-encoder: sum loss -> gradients add, no 1/world
    ep = O.init_encoder_params(1, len(dil), 2, 8, S, 3, bias_scale=0.1)
    dpar = O.init_stack_params(2, dil, 2, R, S, 4 * M, cond_channels=3, bias_scale=0.1)

    def ae_grads(sl):
        te, td = OT.TorchEncoder(ep), OT.TorchStack(dpar)
        loss, _, _ = OT.autoencoder_loss(te, td, torch.tensor(truth[sl][:, :256]), 32)
        loss.backward()
        return torch.cat([v.grad.reshape(-1) for _, v in te.named() if v.grad is not None] +
                         [v.grad.reshape(-1) for _, v in td.named(include_cond=True) if v.grad is not None])

    g = ae_grads(dp.shard_batch(GB, rank, world))
    dp.allreduce_sum_(g)
    assert torch.allclose(g, ae_grads(slice(0, GB)), rtol=1e-9, atol=1e-11)
    dist.barrier()
    dist.destroy_process_group()
    out.put(rank)


def test_dp_student_and_autoencoder_semantics_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_student_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]
