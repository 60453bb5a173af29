#!/usr/bin/env python3
"""How much would 16-bit per-workgroup weight-gradient partials cost in accuracy?  Config 2 (8 x 16000, bf16): the fp32 partial
slabs the backward group kernels leave (one per workgroup and layer) are rounded to bf16 and summed, against their exact
sum: relative L2 of the resulting gradient per tensor and for the worst layer; beside it the error the bf16 mode already
carries against the exact-fp32 mode of the same engine (tests/test_gpu_fullsize.py measures 6-7e-3)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SRWN_PART16"] = "0"      # (this tool reads / stamps the fp32-partial instantiation)
import numpy as np, torch
EG = importlib.import_module("sr-wavenet_amd.engine"); KN = importlib.import_module("sr-wavenet_amd.kernels")
import bench
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
audio = torch.tensor(bench.synthetic_audio(8, 16000, 0), device="cuda")
for steps in (0, 200):
    cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
    eng = EG.WaveNetEngine(cfg, 8, 16000, "cuda", seed=0)
    eng.set_inputs(audio, KN.mu_law_encode(audio, 256))
    for _ in range(steps):
        eng.train_step()
    eng.forward(); eng.backward(); torch.cuda.synchronize()
    L, ns, R = eng.L, eng.nslabs, eng.R
    print("after %d training steps (loss %.4f):" % (steps, float(eng.loss.item())))
    for name, buf, n in (("dWf", eng.pl_f, 2 * R * R), ("dWr", eng.pl_r, R * R), ("dbf", eng.pl_bf, R), ("dbr", eng.pl_br, R)):
        p = buf[:L * ns * n].view(L, ns, n)
        exact = p.double().sum(1)
        for fmt, q in (("bf16", p.bfloat16()), ("fp16", p.half())):
            approx = q.double().sum(1)
            live = exact.norm(dim=1) > 0
            err = float((approx - exact).norm() / exact.norm())
            worst = float(((approx - exact)[live].norm(dim=1) / exact[live].norm(dim=1)).max())
            # per-entry: the entries Adam cares about individually
            rel = ((approx - exact).abs() / exact.abs().clamp_min(1e-30))[live]
            print("  %-4s %s partials: rel L2 %.2e  worst layer %.2e  per-entry median %.2e  99%% %.2e  max|partial| %.3g" %
                  (name, fmt, err, worst, float(rel.median()), float(rel.flatten().kthvalue(int(0.99 * rel.numel())).values), float(p.abs().max())))
