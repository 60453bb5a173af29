#!/usr/bin/env python3
"""In-kernel phase timing of the backward group kernel with the weight gradients inside (srwn_residual_group_bwd_wt;
workgroup 0, waves 0 and 4 = the two waves of SIMD 0): s_memtime cycles between the stamps of one layer."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SRWN_PART16"] = "0"      # (this tool reads / stamps the fp32-partial instantiation)
# the stamped kernel instantiations live in the diagnostic build only (python sr-wavenet_amd/build.py --diag)
os.environ.setdefault("SRWN_LIB_PATH", os.path.join(ROOT, "sr-wavenet_amd", "libsrwn_diag.so"))
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
KN = importlib.import_module("sr-wavenet_amd.kernels")
L = importlib.import_module("sr-wavenet_amd._lib")
B, T = 8, 16000
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
eng = EG.WaveNetEngine(cfg, B, T, "cuda", seed=0)
a = torch.randn(B, T, device="cuda").clamp(-1, 1) * 0.5
eng.set_inputs(a, KN.mu_law_encode(a, 256))
eng.forward(); eng.backward()
torch.cuda.synchronize()
buf = torch.zeros(1024, dtype=torch.int64, device="cuda")
names = {20: "G parked", 21: "barrier", 22: "dWr loop", 23: "barrier", 24: "dWr partial written", 25: "phase A (df)", 26: "barrier",
         27: "phase B (taps -> G)", 28: "dWf loop + partial", 29: "wait for the next weights", 30: "barrier"}
for which in (0, 1):
    g = [g for g in eng.groups if eng.dil[g[0]] == (1 if which == 0 else 32)][-1]
    L.call("srwn_debug_stamp_buffer", buf.data_ptr())
    for _ in range(3):
        buf.zero_()
        eng._group_bwd_wt(g[0], g[1])
        torch.cuda.synchronize()
    L.call("srwn_debug_stamp_buffer", None)
    h = buf.cpu().numpy().astype("uint64")
    for w in (0, 1):
        st = [(int(v) >> 48, int(v) & 0xffffffffffff) for v in h[w * 512:(w + 1) * 512] if v]
        if not st:
            print("wave %d: no stamps" % (4 * w)); continue
        print("---- layers %d..%d (dilations %d..), wave %d: %d stamps, %d cycles from first to last" % (g[0], g[1] - 1, eng.dil[g[0]], 4 * w, len(st), st[-1][1] - st[0][1]))
        agg = {}
        for (t0, c0), (t1, c1) in zip(st[:-1], st[1:]):
            agg.setdefault(t1, []).append(c1 - c0)
        tot = sum(sum(v) for v in agg.values())
        for t, v in sorted(agg.items()):
            print("   -> %-28s n=%2d  mean %7.0f  min %6d  max %6d  share %4.1f %%" % (names.get(t, t), len(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / tot))
