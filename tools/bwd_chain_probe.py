#!/usr/bin/env python3
"""The backward group launches of config 2 timed three ways on one box: span events around the launches in eager steps
(engine.timing), the six launches captured as a hipGraph x 4 and replayed between two events (bench.py's roofline leg),
and the whole step's graph replay for scale."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
EG = importlib.import_module("sr-wavenet_amd.engine"); KN = importlib.import_module("sr-wavenet_amd.kernels")
import bench
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
eng = EG.WaveNetEngine(cfg, 8, 16000, "cuda", seed=0)
audio = torch.tensor(bench.synthetic_audio(8, 16000, 0), device="cuda")
eng.set_inputs(audio, KN.mu_law_encode(audio, 256))
for _ in range(3): eng.train_step()
eng.capture_graphs()
for _ in range(20): eng.train_step_graphed()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): eng.train_step_graphed()
torch.cuda.synchronize()
print("step graph replay: %.4f ms" % ((time.perf_counter() - t0) / 200 * 1e3))
eng.timing = True; eng.spans.clear()
for _ in range(5): eng.train_step()
torch.cuda.synchronize()
v = eng.spans["bwd_layers"]; n = len(v) // 5
print("bwd_layers span (eager, events): %.1f us per launch" % (1e3 * float(np.median([sum(s.elapsed_time(e) for s, e in v[i * n:(i + 1) * n]) for i in range(5)])) / len(eng.groups)))
eng.timing = False
for reps in (1, 4):
    cs = torch.cuda.Stream(); cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        def chain():
            for l0, l1 in reversed(eng.groups): eng._group_bwd_wt(l0, l1)
        chain(); cs.synchronize()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg, stream=cs):
            for _ in range(reps): chain()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cs); cg.replay(); e1.record(cs); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
    torch.cuda.current_stream().wait_stream(cs)
    print("chain graph x%d: %.1f us per launch  (runs: %s)" % (reps, 1e3 * float(np.median(ts)) / (reps * len(eng.groups)), " ".join("%.1f" % (1e3 * t / (reps * len(eng.groups))) for t in ts)))
