#!/usr/bin/env python3
"""Runs only the residual-stack launches of the benchmark configuration (forward layers, data-gradient chain, layer weight
gradients) a few times: the target of `rocprofv3 --pmc ... -- python3 tools/group_probe.py` counter passes."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

EG = importlib.import_module("sr-wavenet_amd.engine")
KN = importlib.import_module("sr-wavenet_amd.kernels")
from oracle import wavenet_np as O   # synthetic input generator only

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B, T = int(os.environ.get("PROBE_B", "8")), int(os.environ.get("PROBE_T", "16000"))
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True,
                     dtype=torch.bfloat16)
eng = EG.WaveNetEngine(cfg, B, T, "cuda", seed=0)
audio = torch.tensor(O.synthetic_audio(B, T, seed=0), device="cuda")
eng.set_inputs(audio, KN.mu_law_encode(audio, 256))
eng.overlap = False
for _ in range(n):
    eng.forward()
    eng.backward()
torch.cuda.synchronize()
print("ok", float(eng.loss.item()))
