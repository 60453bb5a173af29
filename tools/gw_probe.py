#!/usr/bin/env python3
"""In-kernel phase timing of the fused backward + weight-gradient group kernel (csrc/srwn_groupw.hip; workgroup 0,
waves 0 and 1): cycles between stamps, per tag."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
KN = importlib.import_module("sr-wavenet_amd.kernels")
L = importlib.import_module("sr-wavenet_amd._lib")
SA = importlib.import_module("sr-wavenet_amd.simple_audio")
import numpy as np
B, T = 8, 16000
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
eng = EG.WaveNetEngine(cfg, B, T, "cuda", seed=0)
rng = np.random.default_rng(0)
audio = torch.tensor(np.clip(0.5 * np.sin(np.arange(B * T).reshape(B, T) * 0.05) + 0.05 * rng.normal(size=(B, T)), -1, 1), dtype=torch.float32, device="cuda")
eng.set_inputs(audio, KN.mu_law_encode(audio, 256))
eng.forward(); eng.backward()
buf = torch.zeros(1024, dtype=torch.int64, device="cuda")
L.call("srwn_debug_stamp_buffer", buf.data_ptr())
names = {1: "start", 20: "layer start", 21: "A: tile begin (loop overhead)", 22: "A: G staged, gT read, dbr", 23: "A: z staged, zz, cT+gate, dWr mfma",
         24: "A: dcs staged", 25: "A: prefetch issue, chain mfma, dgate, df stored", 26: "A: loop end", 27: "A|B barrier",
         31: "B: tile begin", 32: "B: x staged, frags, dWf mfma, dbf", 33: "B: prefetch issue, chain taps, G packed", 36: "B: loop end", 37: "end barrier", 38: "zero image", 40: "flush: dump", 41: "flush: barrier 1", 42: "flush: reduce + store", 43: "flush: barrier 2"}
for which in (0, 1):
    g = [g for g in eng.groups if eng.dil[g[0]] == (1 if which == 0 else 32)][-1]
    for _ in range(3):
        buf.zero_()
        eng._group_bwd_wg(g[0], g[1])
        torch.cuda.synchronize()
    h = buf.cpu().numpy().astype("uint64")
    for w in (0, 1):
        st = [(int(v) >> 48, int(v) & 0xffffffffffff) for v in h[w * 512:(w + 1) * 512] if v]
        print("---- group dil %d.., wave %d: %d stamps, total %d cycles" % (eng.dil[g[0]], w, len(st), st[-1][1] - st[0][1]))
        agg = {}
        for (t0, c0), (t1, c1) in zip(st[:-1], st[1:]):
            agg.setdefault(t1, []).append(c1 - c0)
        for t, v in sorted(agg.items()):
            print("   -> %-52s n=%3d  mean %7.0f  min %6d  max %6d  sum %8d" % (names.get(t, t), len(v), sum(v) / len(v), min(v), max(v), sum(v)))
L.call("srwn_debug_stamp_buffer", None)
