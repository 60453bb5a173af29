#!/usr/bin/env python3
"""What summing the per-workgroup weight-gradient partials INSIDE the backward group launch would cost:
the six backward launches of config 2 as a hipGraph, timed with HIP events, in the diagnostic library with
SRWN_WT_DEBUG = 0 (as shipped) / 32 (publish: drain + barrier + agent release + ticket per layer) / 96 (publish +
combine by the last of four: 4 x 49 KB read back with sc1 loads).  Results under 32 / 96 are meaningless by design.
usage (GPU box): python tools/fence_probe.py"""
import importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import importlib, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
EG = importlib.import_module("sr-wavenet_amd.engine"); KN = importlib.import_module("sr-wavenet_amd.kernels"); L = importlib.import_module("sr-wavenet_amd._lib")
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
eng = EG.WaveNetEngine(cfg, 8, 16000, "cuda", seed=0)
a = torch.randn(8, 16000, device="cuda").clamp(-1, 1) * 0.5
eng.set_inputs(a, KN.mu_law_encode(a, 256))
eng.forward(); eng.backward(); torch.cuda.synchronize()
buf = torch.zeros(1024 + 4096, dtype=torch.int64, device="cuda")      # 2 x 512 stamps + the ticket counters
if int(os.environ.get("SRWN_WT_DEBUG", "0")) & 32:
    L.call("srwn_debug_stamp_buffer", buf.data_ptr())
def chain():
    for l0, l1 in reversed(eng.groups):
        eng._group_bwd_wt(l0, l1)
chain(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(4): chain()
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) / 24 * 1e3)
print("FENCE %%s %%.2f %%.2f" %% (os.environ.get("SRWN_WT_DEBUG", "0"), float(np.median(ts)), min(ts)))
"""
for rnd in range(2):
    for dbg in (sys.argv[1:] or ["0", "32", "96"]):
        env = dict(os.environ, SRWN_WT_DEBUG=dbg, SRWN_LIB_PATH=os.path.join(ROOT, "sr-wavenet_amd", "libsrwn_diag.so"))
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
        out = [l for l in r.stdout.splitlines() if l.startswith("FENCE")]
        print(out[0] if out else "FAILED %s: %s" % (dbg, r.stderr[-1500:]), flush=True)
