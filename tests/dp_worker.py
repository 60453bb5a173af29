"""One rank of the data-parallel GPU test (tests/test_gpu_dp2.py): a fresh process that shards a fixed global batch,
runs one eager and two graph-replayed training steps of the ENGINE's own data-parallel schedule (gradient all-reduce
in one bucket or two, the first overlapped with the lower backward pass), and saves its parameters.
usage: dp_worker.py <mode: softmax|mol|student|deep> <out.pt>     (RANK / WORLD_SIZE / MASTER_* from the environment)
"deep" is the benchmark's own stack (30 layers 3 x [1..512], bf16) on clips longer than the receptive field, so that the
two-bucket schedule cuts where the benchmark cuts (split_layer = 10, three hipGraphs)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    mode, out = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)            # every rank shares the one GPU of the box; the collective goes through gloo
    import torch.distributed as dist
    # (SRWN_FORCE_DIST=1 with one rank and SRWN_DIST_BACKEND=nccl: the collectives of the schedule run through RCCL -- the
    # only RCCL configuration a one-GPU box allows: two ranks on one card end in "Duplicate GPU detected")
    backend = os.environ.get("SRWN_DIST_BACKEND", "gloo")
    grouped = world > 1 or os.environ.get("SRWN_FORCE_DIST") == "1"
    if grouped:
        kw = {"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    EG = importlib.import_module("sr-wavenet_amd.engine")
    from oracle import wavenet_np as O   # parameter / input generators only
    GB, T, R, S = 4, 700, 64, 256
    dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
    dtype = torch.float32
    if mode == "deep":
        T, dil, dtype = 3200, dil * 3, torch.bfloat16
    b = GB // world
    sl = slice(rank * b, (rank + 1) * b)
    audio = O.synthetic_audio(GB, T, seed=5)
    dev = lambda a, dt=torch.float32: torch.tensor(np.asarray(a), dtype=dt, device="cuda")
    if mode in ("softmax", "mol", "deep"):
        C = 20 if mode == "mol" else 256
        sp = O.init_stack_params(7, dil, 2, R, S, C, bias_scale=0.05)
        cfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, output_channels=C, shift_input=True,
                             dtype=dtype, learning_rate=1e-3,
                             head_mode="mol" if mode == "mol" else "per_timestep")
        eng = EG.WaveNetEngine(cfg, b, T, "cuda")
        eng.load_oracle_params(sp)
        codes = O.mu_law_encode(audio, 256).astype(np.int32)
        eng.set_inputs(dev(audio[sl]), dev(codes[sl], torch.int32))
        info = {"bucketed": bool(eng.bucketed), "world": eng.world, "fused": bool(eng.fused_bwd),
                "split_layer": int(eng.split_layer), "layers": eng.L,
                "backend": dist.get_backend() if grouped else "none"}
        eng.train_step()
        torch.cuda.synchronize()
        grads1, params1 = eng.grads.cpu().clone(), eng.params.cpu().clone()
        eng.capture_graphs()
        info["graphs"] = 1 + (eng._g_b2 is not None) + (eng._g_opt is not None)
        eng.train_step_graphed()
        eng.train_step_graphed()
        torch.cuda.synchronize()
        res = {"params": eng.params.cpu(), "grads1": grads1, "params1": params1, "loss": eng.loss.cpu(), "info": info}
    else:
        ST = importlib.import_module("sr-wavenet_amd.student")
        E, pool, M, F = 5, 70, 5, 2
        rng = np.random.default_rng(11)
        tsp = O.init_stack_params(40, dil, 2, 64, 256, 4 * M, cond_channels=E, bias_scale=0.05)
        flows = [O.init_flow_params(50 + i, dil, 2, R, S, E, bias_scale=0.05) for i in range(F)]
        for p in flows:
            p.head_w2 = p.head_w2 * 0.3
        noise = rng.logistic(0, 1, (GB, T)) * 0.15
        cond = rng.standard_normal((GB, T // pool, E))
        tcfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=4 * M,
                              cond_channels=E, pool_stride=pool, shift_input=True, dtype=torch.float32, head_mode="mol")
        teacher = EG.WaveNetEngine(tcfg, b, T, "cuda"); teacher.load_oracle_params(tsp)
        fcfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, cond_channels=E, pool_stride=pool,
                              dtype=torch.float32)
        stu = ST.StudentEngine(teacher, fcfg, F, alpha=0.8, beta=1.2, gamma=0.05, learning_rate=1e-3)
        for f, p in zip(stu.flows, flows):
            f.load_oracle_params(p)
        stu.set_inputs(dev(noise[sl]), dev(audio[sl]), dev(cond[sl]))
        stu.train_step()
        torch.cuda.synchronize()
        grads1, params1 = stu.storage.grads.cpu().clone(), stu.storage.params.cpu().clone()
        stu.capture_graphs()
        stu.train_step_graphed()
        stu.train_step_graphed()
        torch.cuda.synchronize()
        res = {"params": stu.storage.params.cpu(), "grads1": grads1, "params1": params1,
               "loss": torch.tensor(stu.losses()["loss"]),
               "info": {"world": stu.world}}
    torch.save(res, out)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
