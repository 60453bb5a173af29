#!/usr/bin/env python3
"""Eager steps of one stack shape for rocprofv3 (python3 tools/shape_prof.py R S B T)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
R, S, B, T = (int(v) for v in sys.argv[1:5])
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
cfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, output_channels=256, shift_input=True, dtype=torch.bfloat16)
os.environ["SRWN_OVERLAP"] = "0"
eng = EG.WaveNetEngine(cfg, B, T, "cuda")
eng.audio.copy_(torch.randn(B, T, device="cuda") * 0.3)
eng.targets.copy_(torch.randint(0, 256, (B * T,), device="cuda", dtype=torch.int32))
for _ in range(10):
    eng.train_step()
torch.cuda.synchronize()
