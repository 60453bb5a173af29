#!/usr/bin/env python3
"""Per-step times of the first graph replays after the capture (HIP events between replays): how long the card takes to
reach its steady rate after the host-side capture left it idle."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
EG = importlib.import_module("sr-wavenet_amd.engine"); KN = importlib.import_module("sr-wavenet_amd.kernels")
import bench
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
eng = EG.WaveNetEngine(cfg, 8, 16000, "cuda", seed=0)
audio = torch.tensor(bench.synthetic_audio(8, 16000, 0), device="cuda")
eng.set_inputs(audio, KN.mu_law_encode(audio, 256))
eng.train_step()
t0 = time.perf_counter(); eng.capture_graphs(); torch.cuda.synchronize(); print("capture: %.1f ms of host time" % ((time.perf_counter() - t0) * 1e3))
for idle_ms in (0, 50, 500):
    torch.cuda.synchronize(); time.sleep(idle_ms / 1e3)
    n = 120
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        eng.train_step_graphed(); ev[i + 1].record()
    torch.cuda.synchronize()
    d = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    print("after %3d ms idle: steps 1-5 %s | 6-25 mean %.4f | 26-60 mean %.4f | 61-120 mean %.4f" % (idle_ms, " ".join("%.3f" % x for x in d[:5]), np.mean(d[5:25]), np.mean(d[25:60]), np.mean(d[60:])))

# without events between the replays: windows of K steps, each started after a synchronize (host clock)
for rep in range(2):
    for K in (5, 10, 20, 40, 80, 200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): eng.train_step_graphed()
        torch.cuda.synchronize()
        print("window of %3d steps from a synchronize: %.4f ms/step" % (K, (time.perf_counter() - t0) / K * 1e3))
