#!/usr/bin/env python3
"""The training step's two reduction launches timed alone (HIP events, graph replays of just those launches are not
available: eager calls, median of many): tools/reduce_probe.py [steps]"""
import importlib, os, sys
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
for _ in range(3): eng.train_step()
torch.cuda.synchronize()
calls = []
orig = KN.reduce_partials_multi
def spy(jobs):
    calls.append(list(jobs)); return orig(jobs)
KN.reduce_partials_multi = spy
eng.train_step(); torch.cuda.synchronize()
KN.reduce_partials_multi = orig
print("%d reduction launches per step" % len(calls))
for ci, jobs in enumerate(calls):
    nbytes = 0
    for j in jobs:
        t = j[0]; nbytes += t.numel() * t.element_size() if hasattr(t, "numel") else 0
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(jobs); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print("launch %d: %d jobs, buffers %.1f MB, median %.1f us, min %.1f us" % (ci, len(jobs), nbytes / 1e6, float(np.median(ts)), min(ts)))
    for j in jobs:
        print("    job: nslabs %s n %s batch %s layout %s" % (j[1], j[2], j[3], j[8] if len(j) > 8 else "f32"))
