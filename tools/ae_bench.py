#!/usr/bin/env python3
"""Joint encoder+decoder training step of WaveNetAutoEncoder (the teacher teacher.py trains) on one MI355X.
  --config ref   : teacher.py's own shapes (B=4 x 4096, 32 res / 128 skip ch, pool 512), BASELINE configs[0]
  --config scale : the 30-layer stack at the north-star width (B=8 x 16000, 64 res / 256 skip ch, pool 125)
Not the driver's bench (that is bench.py)."""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="scale")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--graph", type=int, default=1)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
EG = importlib.import_module("sr-wavenet_amd.engine"); EN = importlib.import_module("sr-wavenet_amd.encoder")
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
B, T, R, S, pool, lat, M = (4, 4096, 32, 128, 512, 16, 5) if a.config == "ref" else (8, 16000, 64, 256, 125, 16, 5)
cfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, output_channels=4 * M, cond_channels=lat,
                     pool_stride=pool, shift_input=True, head_mode="mol", dtype=dt, learning_rate=1e-4)
ae = EN.AutoEncoderEngine(cfg, B, T, 128, lat, 0, "cuda")
rng = np.random.default_rng(0)
t = np.arange(T)[None, :]
x = (0.5 * np.sin(2 * np.pi * 110.0 * (1 + np.arange(B))[:, None] * t / 16000) + 0.05 * rng.standard_normal((B, T))).clip(-1, 1)
ae.set_inputs(torch.tensor(x, dtype=torch.float32, device="cuda"))
for _ in range(3):
    ae.train_step()
if a.graph:
    ae.capture_graphs()
step = ae.train_step_graphed if a.graph else ae.train_step
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.steps * 1e3
print(json.dumps({"metric": "audio samples/sec (auto-encoder fwd+bwd+Adam)", "value": B * T / ms * 1e3, "unit": "samples/s",
                  "ms_per_step": ms, "dtype": a.dtype, "loss": float(ae.loss.item()),
                  "config": {"workload": "WaveNetAutoEncoder %s: 30+1 NC layers x128 ch, decoder 30 layers %d/%d, batch %dx%d" % (a.config, R, S, B, T)}}))
