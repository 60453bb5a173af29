#!/usr/bin/env python3
"""Distillation step of BASELINE config 4 on one MI355X: 4 IAF flows x 30 layers (R=64), 10-component
mixture-of-logistics teacher (30 layers, frozen, forward only), batch 8 x 16000, bf16.

Prints one JSON line (audio samples/s through teacher-forward + student fwd/bwd + clipped Adam) and, with
--spans, the HIP-event time of each phase.  Not the driver's bench (that is bench.py, config 2)."""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--flows", type=int, default=4)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--length", type=int, default=16000)
ap.add_argument("--graph", type=int, default=1)
ap.add_argument("--trace", type=int, default=0, help="print the losses every N steps (a synchronisation each: not for timing)")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--phases", type=int, default=0, help="time teacher fwd / flows fwd / losses / backward / update eagerly")
a = ap.parse_args()

EG = importlib.import_module("sr-wavenet_amd.engine")
ST = importlib.import_module("sr-wavenet_amd.student")
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
B, T, pool, lat = a.batch, a.length, 125, 16
M = 10
tcfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=4 * M, cond_channels=lat,
                      pool_stride=pool, shift_input=True, dtype=dt, head_mode="mol")
teacher = EG.WaveNetEngine(tcfg, B, T, "cuda")
fcfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, cond_channels=lat, pool_stride=pool, dtype=dt)
stu = ST.StudentEngine(teacher, fcfg, a.flows, alpha=1.0, beta=1.0, gamma=1e-3, learning_rate=1e-4)
rng = np.random.default_rng(0)
t = np.arange(T)[None, :]
truth = (0.5 * np.sin(2 * np.pi * 110.0 * (1 + np.arange(B))[:, None] * t / 16000) + 0.05 * rng.standard_normal((B, T))).clip(-1, 1)
noise = rng.logistic(0, 1, (B, T))
enc = rng.standard_normal((B, T // pool, lat))
dev = lambda x: torch.tensor(x, dtype=torch.float32, device="cuda")
stu.set_inputs(dev(noise), dev(truth), dev(enc))
for _ in range(max(a.warmup, 1)):
    stu.train_step()
if a.graph:
    stu.capture_graphs()
step = stu.train_step_graphed if a.graph else stu.train_step
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    step()
    if a.trace and (i % a.trace == 0 or i == a.steps - 1):
        l = stu.losses()
        print("step %4d  loss %.6g  entropy %.6g  power %.6g  finite params %s" % (
            i, l["loss"], l["entropy"], l["power_loss"], all(bool(torch.isfinite(f.params).all()) for f in stu.flows)), flush=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.steps * 1e3
l = stu.losses()
out = {"metric": "audio samples/sec (teacher fwd + 4-flow student fwd+bwd+clipped Adam)", "value": B * T / ms * 1e3,
       "unit": "samples/s", "n_gpus": 1, "steps": a.steps, "ms_per_step": ms, "dtype": a.dtype,
       "config": {"workload": "BASELINE configs[3]: %d flows x 30 layers R=64, teacher 30 layers MoL-10, batch %dx%d" % (a.flows, B, T)},
       "loss": l["loss"], "power_loss": l["power_loss"], "entropy": l["entropy"]}
if a.phases:
    def timed(fn, n=5):
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n
    ph = {"teacher_fwd_ms": timed(lambda: teacher.forward(with_loss=False)),
          "flows_fwd_ms": timed(stu.forward_flows), "forward_all_ms": timed(stu.forward)}
    stu.forward()
    ph["backward_ms"] = timed(stu.backward)
    ph["update_ms"] = timed(stu.optimizer_step)
    out["phases_eager"] = ph
print(json.dumps(out))
