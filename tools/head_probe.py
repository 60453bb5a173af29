#!/usr/bin/env python3
"""In-kernel phase timing of the one-launch head (workgroup 0, waves 0 and 1): cycles between stamps."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the stamped kernel instantiations live in the diagnostic build only (python sr-wavenet_amd/build.py --diag)
os.environ.setdefault("SRWN_LIB_PATH", os.path.join(ROOT, "sr-wavenet_amd", "libsrwn_diag.so"))
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
KN = importlib.import_module("sr-wavenet_amd.kernels")
L = importlib.import_module("sr-wavenet_amd._lib")
from oracle import wavenet_np as O
B, T = 8, 16000
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
eng = EG.WaveNetEngine(cfg, B, T, "cuda", seed=0)
audio = torch.tensor(O.synthetic_audio(B, T, seed=0), device="cuda")
eng.set_inputs(audio, KN.mu_law_encode(audio, 256))
eng.forward()
buf = torch.zeros(1024, dtype=torch.int64, device="cuda")
L.call("srwn_debug_stamp_buffer", buf.data_ptr())
names = {1: "start", 2: "prologue", 10: "dma issued", 11: "32 mfma", 12: "dma wait", 13: "epilogue", 14: "barrier"}
def head():       # engine.forward's head launch alone (the stamped forward groups and skip sum would share the buffer)
    v = eng.view
    KN.head_chain(eng.r0, eng.wptr(eng.o_w1), eng.wptr(eng.o_w2p), eng.wptr(eng.o_w2Tp), eng.wptr(eng.o_w1Tp), v("head_b1"), v("head_b2"),
                  eng.targets, eng.loss_parts, eng.r1, eng.dlogits, eng.da1, eng.dtotal, eng.C, 1.0 / eng.N)
for _ in range(3):
    buf.zero_()
    head()
    torch.cuda.synchronize()
h = buf.cpu().numpy().astype("uint64")
for w in (0, 1):
    st = [(int(v) >> 48, int(v) & 0xffffffffffff) for v in h[w * 512:(w + 1) * 512] if v]
    print("---- wave %d: %d stamps, total %d cycles" % (w, len(st), st[-1][1] - st[0][1]))
    agg = {}
    for (t0, c0), (t1, c1) in zip(st[:-1], st[1:]):
        agg.setdefault(t1, []).append(c1 - c0)
    for t, v in sorted(agg.items()):
        print("   -> %-12s n=%2d  mean %7.0f  min %6d  max %6d  sum %7d   %s" % (names.get(t, t), len(v), sum(v) / len(v), min(v), max(v), sum(v), v if len(v) <= 16 else ""))
L.call("srwn_debug_stamp_buffer", None)
