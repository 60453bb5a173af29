import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
EG = importlib.import_module("sr-wavenet_amd.engine"); K = importlib.import_module("sr-wavenet_amd.kernels")
from tools.kbench import timeit
for T in (2000, 4000, 16000, 64000):
    cfg = EG.StackConfig(dilations=[1, 2, 4], dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True)
    eng = EG.WaveNetEngine(cfg, 8, T, "cuda")
    eng.audio.copy_(torch.randn(8, T, device="cuda") * 0.3); eng.forward(); eng.backward()
    us = timeit(lambda: eng._layer_fwd(1, None), reps=50)
    B, R, S = 8, 64, 256
    usb = timeit(lambda: K.residual_layer_bwd(None, eng.dfs[2], eng.wptr(eng.o_convT[2]), eng.gs[2], eng.wptr(eng.o_resT[1]),
                 eng.wptr(eng.o_skipT[1]), eng.dtotal, eng.zs[1], eng.dfs[1], B, T, R, S, 2, eng.dil[2], True, True, torch.bfloat16), reps=50)
    print("T=%6d tiles=%6d  layer_fwd %.1f us (%.2f ns/tile)   layer_bwd %.1f us" % (T, 8 * T // 32, us, us * 1e3 / (8 * T / 32), usb))
