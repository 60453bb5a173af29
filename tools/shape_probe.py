#!/usr/bin/env python3
"""Step time of the teacher stack at other widths (the reference's scripts use 32 residual / 128 skip channels)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
for (R, S, C, B, T) in ((64, 256, 256, 8, 16000), (32, 128, 256, 8, 16000), (32, 256, 256, 8, 16000), (32, 128, 256, 4, 4096)):
    cfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, output_channels=C, shift_input=True,
                         dtype=torch.bfloat16)
    eng = EG.WaveNetEngine(cfg, B, T, "cuda")
    eng.audio.copy_(torch.randn(B, T, device="cuda") * 0.3)
    eng.targets.copy_(torch.randint(0, C, (B * T,), device="cuda", dtype=torch.int32))
    for _ in range(3):
        eng.train_step()
    eng.capture_graphs()
    for _ in range(3):
        eng.train_step_graphed()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        eng.train_step_graphed()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print("R=%d S=%d C=%d B=%d T=%d: %.3f ms/step = %.1f M samples/s" % (R, S, C, B, T, ms, B * T / ms / 1e3), flush=True)
    del eng
    torch.cuda.empty_cache()
