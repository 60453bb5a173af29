#!/usr/bin/env python3
"""One large shape (16 clips of 64000 samples = 1 M rows per step): robustness + throughput away from config 2."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
B, T = 16, 64000
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
eng = EG.WaveNetEngine(cfg, B, T, "cuda")
eng.audio.copy_(torch.randn(B, T, device="cuda") * 0.3)
eng.targets.copy_(torch.randint(0, 256, (B * T,), device="cuda", dtype=torch.int32))
for _ in range(2):
    eng.train_step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    eng.train_step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
print("B=%d T=%d: %.2f ms/step = %.1f M samples/s, loss %.4f, mem %.1f GB" % (B, T, ms, B * T / ms / 1e3, float(eng.loss.item()), torch.cuda.max_memory_allocated() / 2**30))
