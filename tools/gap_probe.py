"""Do launches of one kernel chained in a hipGraph cost more than the kernel alone?  Replays graphs of n = 1, 2, 4, 8
launches of (a) the backward group kernel of layers 5..9, (b) the forward group kernel of the same layers, (c) the skip-sum
GEMM, and prints the time per launch: a gap between dependent nodes shows as a per-launch time that grows with n."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
eng = EG.WaveNetEngine(cfg, 8, 16000, "cuda")
a = torch.randn(8, 16000, device="cuda").clamp(-1, 1) * 0.5
eng.set_inputs(a, torch.randint(0, 256, (8, 16000), dtype=torch.int32, device="cuda"))
eng.forward(); eng.backward()
torch.cuda.synchronize()
l0, l1 = eng.groups[1]
cases = {"group_bwd": lambda: eng._group_bwd_wt(l0, l1), "group_fwd": lambda: eng._group_fwd(l0, l1, None),
         "group_bwd_g0": lambda: eng._group_bwd_wt(*eng.groups[0])}
for name, fn in cases.items():
    for n in (1, 2, 4, 8):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        print("%-14s n=%d  %.1f us per launch" % (name, n, (time.perf_counter() - t0) / 20 / n * 1e6), flush=True)
