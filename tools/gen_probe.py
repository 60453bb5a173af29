"""Per-layer and per-step-intercept cost of the generation kernels: times `nsteps` sampled steps of one workgroup (32
streams) for stacks of 10, 20 and 30 layers; slope = one layer, intercept = input conv + head + sampling.
  python tools/gen_probe.py [nsteps]            (SRWN_GEN16=0: the throughput kernel)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

EG = importlib.import_module("sr-wavenet_amd.engine")
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 512
res = {}
for L in (10, 20, 30):
    dil = ([1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3)[:L]
    cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True,
                         dtype=torch.bfloat16)
    eng = EG.WaveNetEngine(cfg, 1, 64, "cuda")
    for B in (32, 2048):
        eng.generate(64, batch=B)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            eng.generate(nsteps, mode="sample", seed=1, batch=B)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / nsteps * 1e6)
        res[(L, B)] = best
        print("layers %2d streams %4d: %7.2f us per sample" % (L, B, best), flush=True)
for B in (32, 2048):
    slope = (res[(30, B)] - res[(10, B)]) / 20
    print("streams %4d: %.3f us per layer, %.2f us per step outside the layers" % (B, slope, res[(10, B)] - 10 * slope))
