#!/usr/bin/env python3
"""Group kernels over batch sizes 1..8 (T = 16000): how much of a launch is the per-workgroup critical path (does not
shrink with fewer busy CUs) and how much is contention for the memory system (grows with the number of busy CUs)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
KN = importlib.import_module("sr-wavenet_amd.kernels")
from oracle import wavenet_np as O
T = 16000
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3)
    return best
for B in (1, 2, 4, 8):
    cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
    eng = EG.WaveNetEngine(cfg, B, T, "cuda", seed=0)
    audio = torch.tensor(O.synthetic_audio(B, T, seed=0), device="cuda")
    eng.set_inputs(audio, KN.mu_law_encode(audio, 256))
    eng.overlap = False
    eng.forward(); eng.backward(); torch.cuda.synchronize()
    g1, g32 = eng.groups[2], eng.groups[3]
    print("B=%d  fwd st1 %.1f st32 %.1f | bwd st1 %.1f st32 %.1f | wgrad_layers %.1f us" % (
        B, timeit(lambda: eng._group_fwd(g1[0], g1[1], None)), timeit(lambda: eng._group_fwd(g32[0], g32[1], None)),
        timeit(lambda: eng._group_bwd(*g1)), timeit(lambda: eng._group_bwd(*g32)), timeit(lambda: eng._wgrad_layers_group(*g1))))
    del eng; torch.cuda.empty_cache()
