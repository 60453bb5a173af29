#!/usr/bin/env python3
"""Headline benchmark: audio samples/s (fwd+bwd+Adam) of the 30-layer teacher WaveNet on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N>1 under torch.distributed.run, one
rank per GPU over RCCL).  W untimed warm-up steps, then exactly K steps timed between
barrier + torch.cuda.synchronize() on both sides; MAX over ranks; rank 0 prints ONE JSON line.

Workload = BASELINE.json configs[1]: 3x[1..512] dilations, 64 residual / 256 skip channels,
256-way mu-law softmax, batch 8 x 16000 samples per GPU, bf16 MFMA with fp32 accumulation,
synthetic 16 kHz input, random-init weights.  Data parallel = weak scaling.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0        # HBM3E peak (same guide); ~4.9 TB/s is what a plain copy kernel reaches (tools/micro/membench.hip)


def synthetic_audio(B, T, seed=0):
    """x[b,t] = 0.5 sin(2 pi f_b t / 16000) + 0.05 N(0,1), f_b = 110 (b+1) Hz, clipped to [-1, 1], fp32."""
    rng = np.random.default_rng(seed)
    t = np.arange(T, dtype=np.float64)[None, :]
    f = 110.0 * (1 + np.arange(B, dtype=np.float64))[:, None]
    x = 0.5 * np.sin(2.0 * np.pi * f * t / 16000.0) + 0.05 * rng.standard_normal((B, T))
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def profiled_traffic(kernel_substr, config_key):
    """HBM bytes per launch of a kernel from the newest tracked rocprofv3 PMC table under profiles/ whose first line
    names this configuration (tools/hbm_traffic.py writes the table; FETCH_SIZE x 2 + WRITE_SIZE, separate passes):
    launch-weighted mean over the table rows whose kernel name contains `kernel_substr`.  (None, None) if there is none."""
    import glob
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.md")), reverse=True):
        txt = open(path).read()
        if config_key not in txt.split("\n", 1)[0]:
            continue
        tot, calls = 0.0, 0
        for m in re.finditer(r"^\| `([^`]*)` \| (\d+) \| ([0-9.]+) \| ([0-9.]+) \|", txt, flags=re.M):
            if kernel_substr in m.group(1):
                n = int(m.group(2))
                tot += n * (float(m.group(3)) + float(m.group(4))) * 1e6
                calls += n
        if calls:
            return tot / calls, os.path.relpath(path, ROOT)
    return None, None


def algorithmic_flops(N, L, R, S, C, Kw):
    """forward flop/sample of the reference-faithful graph; x3 for fwd+bwd."""
    per_layer = 2 * Kw * R * R + 2 * R * R + 2 * R * S
    fwd = 2 * Kw * R + L * per_layer + 2 * S * S + 2 * S * C
    return dict(per_sample_fwd=fwd, per_sample_fwd_bwd=3 * fwd, per_step=3 * fwd * N)


def cpu_baseline(dil, R, S, C, threads, seconds_budget=20.0):
    """Times the CPU restatement (oracle ii) on a bounded sample of the same workload."""
    from oracle import wavenet_np as O
    from oracle import wavenet_torch as OT
    B, T = 2, 4000
    sp = O.init_stack_params(0, dil, 2, R, S, C)
    audio = O.synthetic_audio(B, T, seed=0)
    codes = O.mu_law_encode(audio, C).astype(np.int64)
    r = OT.cpu_train_steps(sp, audio, codes, steps=1, threads=threads)
    steps = int(max(1, min(20, seconds_budget / max(r["seconds"], 1e-3))))
    if steps > 1:
        r = OT.cpu_train_steps(sp, audio, codes, steps=steps, threads=threads)
    return {"value": r["samples_per_s"], "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": "CPU restatement of the reference graph (oracle/wavenet_torch.py; TensorFlow unavailable), "
                      "fp32 fwd+bwd+Adam, %d steps of batch %dx%d samples (same 30-layer stack)" % (r["steps"], B, T)}


def student_leg(steps=10, warmup=3, B=8, T=16000):
    """BASELINE configs[3] at its one-GPU shape: 4 IAF flows x 30 layers (R = 64) distilled against a frozen 30-layer
    mixture-of-logistics-10 teacher, batch 8 x 16000, bf16; teacher forward + student forward/backward + clipped Adam
    as hipGraphs (student.py:70-107, model.py:290-401).  Returns ms per step."""
    EG = importlib.import_module("sr-wavenet_amd.engine")
    ST = importlib.import_module("sr-wavenet_amd.student")
    dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
    pool, lat, M = 125, 16, 10
    dt = torch.bfloat16
    tcfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=4 * M, cond_channels=lat,
                          pool_stride=pool, shift_input=True, dtype=dt, head_mode="mol")
    teacher = EG.WaveNetEngine(tcfg, B, T, "cuda", frozen=True)   # never trained here: no tiles / partial slabs allocated
    fcfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, cond_channels=lat, pool_stride=pool, dtype=dt)
    stu = ST.StudentEngine(teacher, fcfg, 4, alpha=1.0, beta=1.0, gamma=1e-3, learning_rate=1e-4)
    rng = np.random.default_rng(0)
    dev = lambda x: torch.tensor(x, dtype=torch.float32, device="cuda")
    stu.set_inputs(dev(rng.logistic(0, 1, (B, T))), dev(synthetic_audio(B, T, 0)), dev(rng.standard_normal((B, T // pool, lat))))
    for _ in range(max(warmup, 1)):
        stu.train_step()
    stu.capture_graphs()
    for _ in range(2):
        stu.train_step_graphed()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        stu.train_step_graphed()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def generation_leg(nsteps=256):
    """BASELINE configs[4]: queue-cached autoregressive sampling from the 30-layer mu-law teacher (generator.py; the
    reference runs teacher.py:140-171's whole-clip pass per sample), bf16: microseconds per 16 kHz sample with ONE
    workgroup of 32 streams (the latency of a single stream) and with 2048 streams (the chip's aggregate rate)."""
    EG = importlib.import_module("sr-wavenet_amd.engine")
    dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
    cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True,
                         dtype=torch.bfloat16)
    eng = EG.WaveNetEngine(cfg, 1, 64, "cuda")
    out = {}
    for name, B in (("1wg", 32), ("2048", 2048)):
        eng.generate(64, batch=B)
        torch.cuda.synchronize()
        best = None
        for _ in range(3):      # (the call allocates its rings -- 0.8 GB for 2048 streams -- on the way: a pass that misses the
            t0 = time.perf_counter()      # caching allocator pays a hipMalloc of that size; the fastest of three is the kernel)
            eng.generate(nsteps, mode="sample", seed=1, batch=B)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / nsteps * 1e6
            best = dt if best is None else min(best, dt)
        out[name] = best
    return out


def main():
    # Exactly ONE line goes to stdout: the JSON.  Everything else that may write there on the way (RCCL prints a version
    # banner on stdout when its communicator is created) is sent to stderr: fd 1 is pointed at fd 2 until the final print.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--length", type=int, default=16000)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the student (config 4) and generation (config 5) legs")
    ap.add_argument("--no-graph-launch-timing", action="store_true",
                    help="time the dominant kernel's launches from the eager span events only")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("SRWN_GRAPH", "1")))
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    ndev = torch.cuda.device_count()
    backend = os.environ.get("SRWN_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
    if world > 1 and backend == "nccl" and local >= ndev:
        raise SystemExit("LOCAL_RANK %d but %d visible GPUs: RCCL needs one distinct device per local rank" % (local, ndev))
    local = local % max(ndev, 1)   # (rehearsal: several ranks may share one GPU with SRWN_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("SRWN_FORCE_DIST") == "1":   # (FORCE_DIST: exercise the RCCL path with one rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        # what the line below reports is only a scaling figure if the job is what --gpus says it is
        if dist.get_world_size() != max(args.gpus, 1) and world > 1:
            raise SystemExit("process group has %d ranks but --gpus %d" % (dist.get_world_size(), args.gpus))
        if backend == "nccl" and world > 1:     # one distinct device per rank of this node (a wrapped LOCAL_RANK would
            mine = torch.tensor([local], device="cuda", dtype=torch.int64)      # put two ranks on one card)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            ids = [int(t.item()) for t in every]
            if len(set(ids)) != world:
                raise SystemExit("ranks share devices: %s" % ids)

    EG = importlib.import_module("sr-wavenet_amd.engine")
    KN = importlib.import_module("sr-wavenet_amd.kernels")

    dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
    R, S, C, Kw = 64, 256, 256, 2
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, output_channels=C, shift_input=True,
                         dtype=dt, learning_rate=1e-3)
    eng = EG.WaveNetEngine(cfg, args.batch, args.length, "cuda", seed=0)   # same init on every rank
    B, T, N, L = eng.B, eng.T, eng.N, eng.L
    audio = torch.tensor(synthetic_audio(B, T, seed=rank), device="cuda")
    codes = KN.mu_law_encode(audio, C)
    eng.set_inputs(audio, codes)

    use_graph = bool(args.graph)
    step_fn = eng.train_step
    if use_graph:
        # one eager step (allocations, lazy initialisation), the capture, THEN the W warm-up steps as replays of the
        # captured step: the capture is tens of ms of host work with the GPU idle, and warm-up steps taken before it
        # leave the timed region to start on a card that has clocked down (20 timed steps are 33 ms)
        eng.train_step()
        try:   # {fwd,bwd} and {Adam,pack} as two hipGraphs; the RCCL all-reduce stays between them
            eng.capture_graphs()
            step_fn = eng.train_step_graphed
            step_fn()
        except Exception as e:   # fall back to eager launches, and say so
            print("hipGraph capture failed (%s); running eager" % e, file=sys.stderr)
            use_graph, step_fn = False, eng.train_step
    for _ in range(args.warmup):
        step_fn()
    torch.cuda.synchronize()

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_fn()
    sync()
    dt_s = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt_s], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_s = float(tt.item())
    loss = float(eng.loss.item())

    # ---- outside the timed region, not `value`: the same step over a longer window.  The K timed steps (33 ms at K = 20)
    # start a few ms after the card was idle (graph capture, barrier); A/Bs on one box (tools/ab_step.py: 200-step
    # windows after 20 warm-up replays) read 3-4 % lower than a 20-step window there, so both are reported.
    ms_steady = None
    if use_graph and not args.no_extras:
        nst = 200
        sync()
        t1 = time.perf_counter()
        for _ in range(nst):
            step_fn()
        sync()
        ms_steady = (time.perf_counter() - t1) / nst * 1e3

    # ---- multi-rank only, outside the timed region: how much of the gradient all-reduce the schedule leaves EXPOSED --
    # the time the launch stream spends in the collectives after the lower backward graph has finished (all of bucket B
    # plus whatever tail of bucket A the lower backward did not cover); with one bucket, the whole all-reduce.  HIP
    # events on the launch stream around the collective calls of a few extra replayed steps.
    graphs_per_step, allreduce_exposed_us, allreduce_bytes = None, None, None
    if use_graph:
        graphs_per_step = 1 + (eng._g_b2 is not None) + (eng._g_opt is not None)
    if dist is not None and use_graph and eng._g_opt is not None:
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            eng._g_fb.replay()
            if eng._g_b2 is not None:
                h = eng._allreduce_bucket_a()
                eng._g_b2.replay()
                e0.record()
                eng._allreduce_bucket_b(h)
                e1.record()
            else:
                e0.record()
                eng.allreduce_grads()
                e1.record()
            eng._g_opt.replay()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        tt = torch.tensor([float(np.median(ts))], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        allreduce_exposed_us = float(tt.item())
        allreduce_bytes = {"bucket_a_overlapped": 4 * (eng.nparams - eng.bucket_off) if eng._g_b2 is not None else 0,
                           "bucket_b_exposed": 4 * (eng.bucket_off if eng._g_b2 is not None else eng.nparams)}

    # ---- roofline of the dominant kernel: HIP events on the launch stream, eager launches
    eng.timing = True
    eng.spans.clear()
    npass = max(min(args.steps, 5), 1)
    for _ in range(npass):
        eng.train_step()
    torch.cuda.synchronize()
    eng.timing = False
    # second timing pass: the same spans with the side-stream overlap of the real schedule left on (only where the
    # schedule has one: with the group kernels the default schedule runs one kernel at a time, SRWN_OVERLAP)
    spans_serial = dict(eng.spans)
    spans_ov_raw = spans_serial
    if eng.overlap:
        eng.spans = {}
        eng.timing_overlap = True
        for _ in range(npass):
            eng.train_step()
        torch.cuda.synchronize()
        eng.timing_overlap = False
        spans_ov_raw, eng.spans = dict(eng.spans), spans_serial
    # median over the timing passes (a mean lets one stray pass -- a first-touch allocation, a late module load --
    # stand for the kernel: round 1's driver record carried a 25x outlier that way), summed over the launches of a span
    def _span_ms(v):
        per_pass = len(v) // npass if len(v) >= npass else 1
        tot = [sum(s.elapsed_time(e) for s, e in v[i * per_pass:(i + 1) * per_pass]) for i in range(len(v) // per_pass)]
        return float(np.median(tot))
    spans = {k: _span_ms(v) for k, v in eng.spans.items()}   # ms per step
    spans_ov = {k: _span_ms(v) for k, v in spans_ov_raw.items()}
    fl = algorithmic_flops(N, L, R, S, C, Kw)
    kflops = {"skip_sum": 2.0 * N * L * R * S, "wgrad_skip": 2.0 * N * L * R * S}
    es = 2 if args.dtype == "bf16" else 4
    step_ms = 1e3 * dt_s / args.steps
    # Dominant kernel by time = the backward chain of the residual stack, HBM-bound.
    #  default path (fused_wt): group_bwd_kernel in its weight-gradient-tile instantiation, one launch per layer group.
    #    Algorithmic bytes per launch: per sample and layer it reads z, dcs and the transposed x and c tiles (4 x R x es
    #    bytes), reads the group's top gradient (where there is one) and writes its bottom gradient once (R x es each);
    #    the fp32 partial sums it leaves (one 3 R R + 2 R block per workgroup and layer) are in the measured traffic.
    #  SRWN_FUSE_WT=0: the same kernel without the weight gradients: reads z, dcs, writes df, G (4 x R x es) + top gradient;
    #  SRWN_FUSE=0: layer_bwd_kernel, one launch per layer: reads G_{l+2}, df_{l+1}, dcs_l, z_l, writes G_{l+1}, df_l.
    fused = eng.fused_bwd
    cfg_key = "config: %s B=%d T=%d L=%d R=%d S=%d" % (args.dtype, B, T, L, R, S)
    # The operand list behind bytes_per_launch is stated in the line (roofline.operands), and the same figure over the
    # MINIMAL operand set beside it (frac_min_operands): c^T is z sigmoid(z) stored a second time in another layout, so
    # counting it raises the algorithmic bytes without being information the kernel could not have derived.
    if fused:
        ngr = len(eng.groups)
        extra = 1.0 if eng.fused_wt else 0.0      # (the bottom gradient; without the weight gradients it is one of the 4)
        bwd_bytes_step = sum((4.0 * (l1 - l0) + extra + (1.0 if l1 < L else 0.0)) * R * es * N for l0, l1 in eng.groups)
        nlaunch, kname = ngr, "group_bwd_kernel"
        traffic, traffic_src = profiled_traffic("group_bwd_kernel", cfg_key + (" wt=1" if eng.fused_wt else " wt=0"))
        if eng.fused_wt:
            operands = ("per layer and sample row: z, dcs, x^T tile, c^T tile read (4 x R x %d B); per launch: the group's "
                        "top gradient read (where there is one), its bottom gradient written (R x %d B each); the fp32 "
                        "weight-gradient partials it leaves are in `traffic`, not here" % (es, es))
            min_bytes_step = sum((3.0 * (l1 - l0) + 1.0 + (1.0 if l1 < L else 0.0)) * R * es * N for l0, l1 in eng.groups)
            operands_min = "x, z, dcs per layer (c^T = z sigmoid(z) is derivable from z) + top gradient in + bottom gradient out"
        else:
            operands = "per layer and sample row: z, dcs read, df, G written (4 x R x %d B); per launch: the top gradient read" % es
            min_bytes_step, operands_min = bwd_bytes_step, "the same"
    else:
        bwd_bytes_step = 6.0 * R * es * N * (L + 1)
        nlaunch, kname = L + 1, "layer_bwd_kernel"
        traffic, traffic_src = profiled_traffic("layer_bwd_kernel", cfg_key + " fuse=0")
        operands = "per launch and sample row: G_{l+2}, df_{l+1}, dcs_l, z_l read, G_{l+1}, df_l written (6 x R x %d B)" % es
        min_bytes_step, operands_min = bwd_bytes_step, "the same"
    bwd_launch_ms = spans["bwd_layers"] / nlaunch
    eager_launch_us = 1e3 * bwd_launch_ms
    # The eager passes above put an event pair around every launch (the per-kernel spans), and each event is a packet
    # between two kernels: the launches of the dominant kernel come out 15-20 % longer than the profiler's durations.
    # So the figure the roofline uses is taken the way the timed region runs them: the step's backward launches captured
    # into a hipGraph (four times over: they only read the forward's buffers), replayed between two events on the
    # launch stream.
    if fused and eng.fused_wt and not args.no_graph_launch_timing:
        reps = 4
        cs = torch.cuda.Stream()
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            def chain():
                for l0, l1 in reversed(eng.groups):
                    eng._group_bwd_wt(l0, l1)
            chain()
            cs.synchronize()
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg, stream=cs):
                for _ in range(reps):
                    chain()
            for _ in range(4):      # (untimed: a replay is 2.4 ms, and the first ones after the idle capture run 10 % long)
                cg.replay()
            ts = []
            for _ in range(11):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cs)
                cg.replay()
                e1.record(cs)
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
        torch.cuda.current_stream().wait_stream(cs)
        bwd_launch_ms = float(np.median(ts)) / (reps * nlaunch)
    ach_bw = bwd_bytes_step / nlaunch / (bwd_launch_ms * 1e-3) / 1e9
    roofline = {"kernel": kname, "bound": "hbm", "achieved": ach_bw, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": ach_bw / PEAK_HBM_GBS,
                # HBM bytes per launch from the rocprofv3 PMC passes of the same command (FETCH_SIZE x 2 + WRITE_SIZE,
                # MI355X_MICROARCH.md), read at run time from the tracked profile named in traffic_source (null when
                # no tracked profile covers this configuration); not measured in this run
                "traffic": traffic, "traffic_source": traffic_src,
                "bytes_per_launch": bwd_bytes_step / nlaunch, "operands": operands,
                "bytes_per_launch_min_operands": min_bytes_step / nlaunch, "operands_min": operands_min,
                "frac_min_operands": min_bytes_step / nlaunch / (bwd_launch_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                # what a streaming kernel with this kernel's mix of directions (87 % reads) reaches on this chip
                # (tools/micro/cuingest.hip, profiles/r04_q_cu_ingest.txt: 1 KB written per 4 KB read); not a peak
                "stream_rate_same_mix_GBs": 5370.0,
                "launch_us": 1e3 * bwd_launch_ms,
                "launch_us_eager_with_span_events": eager_launch_us,
                # the same launches timed inside the schedule the timed region replays (weight-gradient passes running
                # beside them on the side stream): what the kernel costs in the step, vs. alone on the chip above
                "launch_us_in_schedule": 1e3 * spans_ov["bwd_layers"] / nlaunch if "bwd_layers" in spans_ov else None,
                "launches_per_step": nlaunch, "share_of_step": nlaunch * bwd_launch_ms / step_ms,
                "whole_step_mfma_frac": fl["per_step"] / (dt_s / args.steps) / 1e12 / PEAK_BF16_TFLOPS,
                "spans_ms": spans}
    # the largest single kernel by FLOPs: the skip sum as one K = L*R contraction (row-streaming MFMA GEMM)
    dom = max((k for k in spans if k in ("skip_sum", "wgrad_skip")), key=lambda k: spans[k])
    achieved = kflops[dom] / (spans[dom] * 1e-3) / 1e12
    roofline_gemm = {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_BF16_TFLOPS,
                     }
    roofline_gemm["traffic"], roofline_gemm["traffic_source"] = profiled_traffic(
        {"skip_sum": "rowgemm_kernel", "wgrad_skip": "wgrad256_kernelIDF16bLi1"}[dom], cfg_key)

    if rank == 0:
        out = {
            "metric": "audio samples/sec (fwd+bwd) 30-layer teacher WaveNet",
            "value": world * N * args.steps / dt_s, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt_s / args.steps, "ms_per_step_200_more_steps": ms_steady, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "30-layer teacher WaveNet (3x[1..512] dilations, 64 res / 256 skip ch, 256-way "
                                   "mu-law softmax), fwd+bwd+Adam, batch %dx%d samples per GPU" % (B, T),
                       "global_batch": world * B, "seq_len": T, "parallelism": "dp%d" % world,
                       "launch": "hipGraph" if use_graph else "eager", "graphs_per_step": graphs_per_step,
                       "final_loss": loss},
            "roofline": roofline,
            "roofline_gemm": roofline_gemm,
        }
        if dist is not None:
            # (not part of `value`: measured on extra steps after the timed region)
            out["allreduce_exposed_us"] = allreduce_exposed_us
            out["allreduce_bytes"] = allreduce_bytes
            out["dist_backend"] = dist.get_backend()
        if not args.no_extras and world == 1 and args.dtype == "bf16":
            # after (and outside) the timed region: BASELINE configs[3] and [4] at their one-GPU shapes
            del eng
            torch.cuda.empty_cache()
            extra = {}
            try:
                extra["student_ms_per_step"] = student_leg()
                extra["student_samples_per_s"] = 8 * 16000 / extra["student_ms_per_step"] * 1e3
                extra["student_config"] = "4 flows x 30 layers (R=64) + frozen 30-layer MoL-10 teacher forward, 8x16000, bf16, hipGraph"
            except Exception as e:      # (the headline line must survive a failure of an extra leg)
                extra["student_error"] = repr(e)
            try:
                g = generation_leg()
                extra["gen_us_per_sample_1wg"] = g["1wg"]
                extra["gen_rtf_per_stream"] = g["1wg"] * 16000 / 1e6          # > 1: slower than real time
                extra["gen_us_per_sample_2048_streams"] = g["2048"]
                extra["gen_aggregate_x_real_time_2048_streams"] = 2048 / (g["2048"] * 16000 / 1e6)
                extra["gen_config"] = "30-layer mu-law teacher, bf16, 256 sampled steps; '1wg' = 32 streams (one ring group)"
            except Exception as e:
                extra["gen_error"] = repr(e)
            out["extra"] = extra
        if not args.no_cpu_baseline and world == 1:
            # the CPU restatement on the box's host cores, and the same on ONE thread (half the time budget)
            out["cpu_baseline"] = cpu_baseline(dil, R, S, C, threads=min(os.cpu_count() or 1, 16), seconds_budget=16.0)
            out["cpu_baseline_1thread"] = cpu_baseline(dil, R, S, C, threads=1, seconds_budget=8.0)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
