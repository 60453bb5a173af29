#!/usr/bin/env python3
"""Does the Infinity Cache serve a consumer kernel that runs right after its producer?  For batch sizes 1..8 times
group_bwd (writes df, G; reads z, dcs) followed by wgrad_layers of the same group (reads x, z, df, G) and reports the
consumer's bytes / time: if the rate rises while producer + consumer footprint drops under 256 MB, it does."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
KN = importlib.import_module("sr-wavenet_amd.kernels")
from oracle import wavenet_np as O
T = 16000
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
for B in (1, 2, 4, 8):
    cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
    eng = EG.WaveNetEngine(cfg, B, T, "cuda", seed=0)
    audio = torch.tensor(O.synthetic_audio(B, T, seed=0), device="cuda")
    eng.set_inputs(audio, KN.mu_law_encode(audio, 256))
    eng.overlap = False
    eng.forward(); eng.backward()
    torch.cuda.synchronize()
    big = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device="cuda")
    def run(flush):
        ts = []
        for g in (eng.groups[3], eng.groups[2]):
            for _ in range(4):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                e[0].record(); eng._group_bwd(*g); e[1].record()
                if flush:
                    big.fill_(1)          # 300 MB of stores between producer and consumer
                e[2].record(); eng._wgrad_layers_group(*g); e[3].record()
                torch.cuda.synchronize()
                ts.append((e[0].elapsed_time(e[1]) * 1e3, e[2].elapsed_time(e[3]) * 1e3))
        return min(t[0] for t in ts), min(t[1] for t in ts)
    N = B * T
    wb = 5 * 4 * 128 * N / 1e6
    a = run(False); b = run(True)
    print("B=%d  group_bwd %.1f us | wgrad_layers right after %.1f us (%.2f TB/s) | after a 300 MB flush %.1f us (%.2f TB/s)"
          % (B, a[0], a[1], wb / a[1], b[1], wb / b[1]))
    del eng, big
    torch.cuda.empty_cache()
