#!/usr/bin/env python3
"""Socket power and shader clock while ONE kernel of the training step runs back to back (rocm-smi samples beside a
launch loop): which kernels sit at the power limit."""
import importlib, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
KN = importlib.import_module("sr-wavenet_amd.kernels")
import numpy as np
B, T = 8, 16000
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
eng = EG.WaveNetEngine(cfg, B, T, "cuda", seed=0)
rng = np.random.default_rng(0)
audio = torch.tensor(np.clip(0.5 * np.sin(2 * np.pi * 110.0 * (1 + np.arange(B))[:, None] * np.arange(T)[None, :] / 16000)
                             + 0.05 * rng.standard_normal((B, T)), -1, 1), dtype=torch.float32, device="cuda")
eng.set_inputs(audio, KN.mu_law_encode(audio, 256))
eng.forward(); eng.backward(); torch.cuda.synchronize()
N, R, S, L = eng.N, eng.R, eng.S, eng.L
g1, g32 = eng.groups[0], eng.groups[1]
work = {
    "group_fwd 1..16": lambda: eng._group_fwd(g1[0], g1[1], None),
    "group_fwd 32..512": lambda: eng._group_fwd(g32[0], g32[1], None),
    "group_bwd 1..16": (lambda: eng._group_bwd_wt(g1[0], g1[1])) if eng.fused_wt else (lambda: eng._group_bwd(g1[0], g1[1])),
    "group_bwd 32..512": (lambda: eng._group_bwd_wt(g32[0], g32[1])) if eng.fused_wt else (lambda: eng._group_bwd(g32[0], g32[1])),
    "skip_sum": lambda: KN.pw_linear(eng.zs.data_ptr(), R, N * R, R, L * R, eng.wptr(eng.o_skip), eng.bs_sum, eng.r0, S, S, N, pro=KN.PRO_GATE, epi=KN.EPI_RELU),
    "wgrad_skip (from the tiles)": (lambda: KN.wgrad_skip_wt(eng.cTs, eng.wt_layer_st, eng.wt_layer_seg, eng.dtotal, eng.wg_parts, eng.wg_bparts, eng.ns_skip_wt, B, T, R)) if getattr(eng, "skip_wt", False) else (lambda: None),
    "wgrad_skip (wgrad256 on z)": lambda: KN.wgrad256(eng.zs.data_ptr(), N * R, R, L, eng.dtotal, eng.wg_parts, eng.wg_bparts, N, eng.ns_skip, pro=KN.PRO_GATE, chunk_width=R),
    "colgemm": lambda: KN.skip_dgrad_all(eng.dtotal, eng.wptr(eng.o_skipT_all), eng.dcs.view(L, N, R), R, S),
    "head_chain": lambda: KN.head_chain(eng.r0, eng.wptr(eng.o_w1), eng.wptr(eng.o_w2p), eng.wptr(eng.o_w2Tp), eng.wptr(eng.o_w1Tp), eng.view("head_b1"), eng.view("head_b2"), eng.targets, eng.loss_parts, eng.r1, eng.dlogits, eng.da1, eng.dtotal, eng.C, 1.0 / N),
    "whole step": lambda: eng.train_step(),
}


def sample():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    w = [l.split(":")[-1].strip() for l in out.splitlines() if "Package Power" in l]
    c = [l.split("(")[-1].rstrip(")") for l in out.splitlines() if "sclk" in l]
    return (w[0] if w else "?"), (c[0] if c else "?")


if not eng.fused_wt:
    work["wgrad_layers"] = lambda: eng._wgrad_layers_group(g1[0], g1[1])
print("| kernel (launched back to back for ~1.5 s) | us/launch | socket power W (3 samples) | sclk (3 samples) |\n|---|---|---|---|")
for name, fn in work.items():
    stop = False
    def loop():
        while not stop:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
    th = threading.Thread(target=loop)
    th.start()
    time.sleep(1.5)
    s = [sample() for _ in range(3)]
    stop = True
    th.join()
    # per-launch time
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record(); torch.cuda.synchronize()
    print("| %s | %.1f | %s | %s |" % (name, e0.elapsed_time(e1) * 20, " / ".join(x[0] for x in s), " / ".join(x[1] for x in s)), flush=True)
