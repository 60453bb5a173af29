#!/usr/bin/env python3
"""Real-time factor of queue-cached generation, BASELINE config 5: 30-layer teacher, 16 kHz, 1 GPU."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
for dt in (torch.bfloat16, torch.float32):
    cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=dt)
    eng = EG.WaveNetEngine(cfg, 1, 64, "cuda")
    for B in (1, 32, 2048):
        n = 4000
        eng.generate(200, batch=B); torch.cuda.synchronize()
        t0 = time.perf_counter(); a, c, _ = eng.generate(n, mode="sample", seed=1, batch=B); torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        print("%s  B=%2d  %d steps in %.3f s -> %.1f us/step, RTF %.3f per stream (16 kHz), aggregate %.1fx real time"
              % (str(dt).split(".")[-1], B, n, dt_s, dt_s / n * 1e6, dt_s / n * 16000, B * n / 16000 / dt_s), flush=True)

# the conditioned mixture-of-logistics decoder of WaveNetAutoEncoder (the teacher generator.py samples from)
pool, lat, Mx = 125, 16, 10
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=4 * Mx, cond_channels=lat,
                     pool_stride=pool, shift_input=True, head_mode="mol", dtype=torch.bfloat16)
eng = EG.WaveNetEngine(cfg, 1, pool, "cuda")
for B in (1, 32, 2048):
    n = 4000
    cond = torch.randn((B, n // pool, lat), device="cuda")
    eng.generate(250, batch=B, cond=cond[:, :2]); torch.cuda.synchronize()
    t0 = time.perf_counter(); a, c, _ = eng.generate(n, mode="sample", seed=1, batch=B, cond=cond); torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    print("bfloat16 MoL-%d decoder, conditioned  B=%2d  %d steps in %.3f s -> %.1f us/step, RTF %.3f per stream, aggregate %.1fx real time"
          % (Mx, B, n, dt_s, dt_s / n * 1e6, dt_s / n * 16000, B * n / 16000 / dt_s), flush=True)

# the reference scripts' own widths (teacher.py: 32 residual / 128 skip channels; generator.py's defaults: 32 / 256), bf16
for R, S in ((32, 128), (32, 256)):
    for gen16 in ("1", "0"):
        os.environ["SRWN_GEN16"] = gen16
        cfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, output_channels=4 * Mx, cond_channels=lat,
                             pool_stride=pool, shift_input=True, head_mode="mol", dtype=torch.bfloat16)
        eng = EG.WaveNetEngine(cfg, 1, pool, "cuda")
        for B in (1, 2048):
            n = 4000
            cond = torch.randn((B, n // pool, lat), device="cuda")
            eng.generate(250, batch=B, cond=cond[:, :2]); torch.cuda.synchronize()
            t0 = time.perf_counter(); eng.generate(n, mode="sample", seed=1, batch=B, cond=cond); torch.cuda.synchronize()
            dt_s = time.perf_counter() - t0
            print("bfloat16 MoL-%d decoder %d/%d channels, %s kernel  B=%4d -> %.1f us/step, RTF %.3f per stream, aggregate %.1fx real time"
                  % (Mx, R, S, "latency" if gen16 == "1" else "throughput", B, dt_s / n * 1e6, dt_s / n * 16000, B * n / 16000 / dt_s), flush=True)
os.environ.pop("SRWN_GEN16", None)
