#!/usr/bin/env python3
"""Times the forward layer launches of the benchmark configuration with HIP events (ablation runs: SRWN_GDBG)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
KN = importlib.import_module("sr-wavenet_amd.kernels")
from oracle import wavenet_np as O
B, T = 8, 16000
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
eng = EG.WaveNetEngine(cfg, B, T, "cuda", seed=0)
audio = torch.tensor(O.synthetic_audio(B, T, seed=0), device="cuda")
eng.set_inputs(audio, KN.mu_law_encode(audio, 256))
eng.overlap = False
eng.forward(); eng.backward()
def timeit(fn, n=20):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
g1 = [g for g in eng.groups if eng.dil[g[0]] == 1][0]
g32 = [g for g in eng.groups if eng.dil[g[0]] == 32][0]
print("GDBG=%s fwd st1 %.1f us  st32 %.1f us | bwd st1 %.1f us st32 %.1f us" % (
    os.environ.get("SRWN_GDBG", "0"),
    timeit(lambda: eng._group_fwd(g1[0], g1[1], None)), timeit(lambda: eng._group_fwd(g32[0], g32[1], None)),
    timeit(lambda: eng._group_bwd(g1[0], g1[1])), timeit(lambda: eng._group_bwd(g32[0], g32[1]))))
