#!/usr/bin/env python3
"""In-kernel phase timing of the skip sum (rowgemm, gate prologue, relu epilogue: workgroup 0, waves 0 and 4 = the two waves of SIMD 0): cycles of the
s_memtime clock between the stamps of one chunk (= one layer's 64 x 256 weights): gate, issue of the next image + loads,
the 32 MFMAs behind their LDS reads, the wait for the image, the barrier."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
eng.forward()
torch.cuda.synchronize()
buf = torch.zeros(1024, dtype=torch.int64, device="cuda")
L.call("srwn_debug_stamp_buffer", buf.data_ptr())
names = {10: "barrier -> chunk start", 11: "activations arrived + gate", 12: "next image + loads issued", 13: "32 MFMAs (LDS reads ahead)",
         14: "wait for the image", 15: "barrier"}
def skip_sum():      # engine.forward's skip-sum launch alone (the other stamped kernels would write into the same buffer)
    KN.pw_linear(eng.zs.data_ptr(), eng.R, eng.N * eng.R, eng.R, eng.L * eng.R, eng.wptr(eng.o_skip), eng.bs_sum, eng.r0, eng.S, eng.S,
                 eng.N, pro=KN.PRO_GATE, epi=KN.EPI_RELU)
for _ in range(3):
    buf.zero_()
    skip_sum()
    torch.cuda.synchronize()
L.call("srwn_debug_stamp_buffer", None)
h = buf.cpu().numpy().astype("uint64")
for w in (0, 1):
    st = [(int(v) >> 48, int(v) & 0xffffffffffff) for v in h[w * 512:(w + 1) * 512] if v]
    if not st:
        print("wave %d: no stamps" % (4 * w)); continue
    print("---- wave %d: %d stamps, %d s_memtime cycles from first to last" % (4 * w, len(st), st[-1][1] - st[0][1]))
    agg = {}
    for (t0, c0), (t1, c1) in zip(st[:-1], st[1:]):
        agg.setdefault(t1, []).append(c1 - c0)
    tot = sum(sum(v) for v in agg.values())
    for t, v in sorted(agg.items()):
        print("   -> %-30s n=%3d  mean %7.1f  min %6d  max %6d  share %4.1f %%" % (names.get(t, t), len(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / tot))
