#!/usr/bin/env python3
"""Per-kernel timing at the config-2 shapes (B=8, T=16000, L=30, R=64, S=256, C=256), HIP events.
Usage: python tools/kbench.py [name-filter ...]   (run on the GPU box)"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

EG = importlib.import_module("sr-wavenet_amd.engine")
K = importlib.import_module("sr-wavenet_amd.kernels")


def timeit(fn, reps=20, warm=3):
    """GPU time per call: `reps` launches captured into one hipGraph (no host launch overhead), replayed 3x."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (3 * reps) * 1e3   # us


def main():
    filt = sys.argv[1:]
    dt = torch.bfloat16
    dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
    cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256,
                         shift_input=True, dtype=dt)
    eng = EG.WaveNetEngine(cfg, 8, 16000, "cuda")
    B, T, N, L, R, S, Cp = eng.B, eng.T, eng.N, eng.L, eng.R, eng.S, eng.Cp
    eng.audio.copy_(torch.randn(B, T, device="cuda") * 0.3)
    eng.targets.copy_(torch.randint(0, 256, (N,), device="cuda", dtype=torch.int32))
    eng.forward(); eng.backward(); torch.cuda.synchronize()
    v = eng.view
    NR = N * R
    cases = {
        "layer_fwd(d=1)": (lambda: eng._layer_fwd(0, None), 2.0 * N * (2 * R * R + R * R), N * R * 2 * 3),
        "layer_fwd(d=512)": (lambda: eng._layer_fwd(9, None), 2.0 * N * (2 * R * R + R * R), N * R * 2 * 3),
        "skip_sum": (lambda: K.pw_linear(eng.zs.data_ptr(), R, N * R, R, L * R, eng.wptr(eng.o_skip), eng.bs_sum,
                                         eng.r0, S, S, N, pro=K.PRO_GATE, epi=K.EPI_RELU),
                     2.0 * N * L * R * S, N * (L * R + S) * 2),
        "skip_sum_nogate": (lambda: K.pw_linear(eng.zs.data_ptr(), R, N * R, R, L * R, eng.wptr(eng.o_skip), eng.bs_sum,
                                                eng.r0, S, S, N, pro=K.PRO_NONE, epi=K.EPI_RELU),
                            2.0 * N * L * R * S, N * (L * R + S) * 2),
        "wgrad256_skip_nogate": (lambda: K.wgrad256(eng.zs.data_ptr(), NR, R, L, eng.dtotal, eng.wg_parts, eng.wg_bparts, N,
                                                    eng.ns_skip, pro=K.PRO_NONE), 2.0 * N * L * R * S, N * L * R * 2 + N * S * 2),
        "head_1x1": (lambda: K.pw_linear(eng.r0.data_ptr(), S, 0, S, S, eng.wptr(eng.o_w1), v("head_b1"), eng.r1, S,
                                         S, N, epi=K.EPI_RELU), 2.0 * N * S * S, N * S * 4),
        "head_ce": (lambda: K.head_softmax_ce(eng.r1, eng.wptr(eng.o_w2), v("head_b2"), eng.targets, eng.loss_parts,
                                              eng.dlogits, None, Cp, eng.C, 1.0 / N), 2.0 * N * S * Cp, N * S * 4),
        "bwd_head_mask": (lambda: K.pw_linear(eng.dlogits.data_ptr(), Cp, 0, Cp, Cp, eng.wptr(eng.o_w2T), None,
                                              eng.da1, S, S, N, aux=eng.r1, epi=K.EPI_MASK),
                          2.0 * N * S * Cp, N * S * 6),
        "layer_bwd(l=10)": (lambda: K.residual_layer_bwd(eng.gs[12], eng.dfs[11], eng.wptr(eng.o_convT[11]),
                                                         eng.gs[11], eng.wptr(eng.o_resT[10]),
                                                         eng.wptr(eng.o_skipT[10]), eng.dtotal, eng.zs[10],
                                                         eng.dfs[10], B, T, R, S, 2, eng.dil[11], True, True, dt),
                            2.0 * N * (2 * R * R + R * R + R * S), N * (R * 2 * 6 + S * 2)),
        "layer_bwd_dcs(l=10)": (lambda: K.residual_layer_bwd(eng.gs[12], eng.dfs[11], eng.wptr(eng.o_convT[11]),
                                                             eng.gs[11], eng.wptr(eng.o_resT[10]), None, None,
                                                             eng.zs[10], eng.dfs[10], B, T, R, S, 2, eng.dil[11],
                                                             True, True, dt, dcs=eng.dcs[10]),
                                2.0 * N * (2 * R * R + R * R), N * R * 2 * 6),
        "skip_dgrad_all": (lambda: K.skip_dgrad_all(eng.dtotal, eng.wptr(eng.o_skipT_all), eng.dcs.view(L, N, R), R, S),
                           2.0 * N * L * R * S, N * (L * R + S) * 2),
        "wgrad_skip(30)": (lambda: K.wgrad(eng.zs.data_ptr(), NR, R, eng.dtotal.data_ptr(), 0, S, None, L,
                                           eng.wg_parts, eng.wg_bparts, N, T, eng.nslabs, dt, pro=K.PRO_GATE),
                           2.0 * N * L * R * S, N * L * R * 2 + N * S * 2),
        "wgrad256_skip": (lambda: K.wgrad256(eng.zs.data_ptr(), NR, R, L, eng.dtotal, eng.wg_parts, eng.wg_bparts, N,
                                             eng.ns_skip, pro=K.PRO_GATE), 2.0 * N * L * R * S, N * L * R * 2 + N * S * 2),
        "wgrad256_head": (lambda: K.wgrad256(eng.r0.data_ptr(), 64, S, S // 64, eng.da1, eng.wg_parts, eng.wg_bparts,
                                             N, eng.ns_head), 2.0 * N * S * S, N * S * 4),
        "wgrad_layers(30)": (lambda: K.wgrad_layers(eng.xs.view(L + 1, N, R), eng.zs.view(L, N, R), eng.dfs.view(L, N, R),
                                                    eng.gs.data_ptr() + NR * 2, eng.dil, eng.pl_f, eng.pl_r, eng.pl_bf,
                                                    eng.pl_br, T, eng.nslabs), 2.0 * N * L * 3 * R * R, N * L * R * 2 * 4),
        "wgrad_conv_tap(30)": (lambda: K.wgrad(eng.xs.data_ptr(), NR, R, eng.dfs.data_ptr(), NR, R, list(eng.dil), L,
                                               eng.wg_parts, None, N, T, eng.nslabs, dt),
                               2.0 * N * L * R * R, N * L * R * 4),
        "wgrad_head(256x256)": (lambda: K.wgrad(eng.r0.data_ptr(), 0, S, eng.da1.data_ptr(), 0, S, None, 1,
                                                eng.wg_parts, eng.wg_bparts, N, T, eng.nslabs, dt),
                                2.0 * N * S * S, N * S * 4),
        "reduce_partials(skip)": (lambda: K.reduce_partials(eng.wg_parts, eng.nslabs, R * S, L, True, 1.0,
                                                            eng.grads.data_ptr() + 4 * eng.sections["WS"].offset,
                                                            R * S), 0, eng.nslabs * L * R * S * 4),
        "adam+pack": (lambda: eng.optimizer_step(), 0, eng.nparams * 16),
        "forward(all)": (lambda: eng.forward(), 0, 0),
        "backward(all)": (lambda: eng.backward(), 0, 0),
        "train_step": (lambda: eng.train_step(), 3 * 1982720.0 * N, 0),
    }
    print("%-24s %10s %10s %10s" % ("kernel", "us", "TFLOP/s", "GB/s(alg)"))
    for name, (fn, fl, by) in cases.items():
        if filt and not any(f in name for f in filt):
            continue
        us = timeit(fn)
        print("%-24s %10.1f %10.1f %10.1f" % (name, us, fl / us / 1e6, by / us / 1e3))


if __name__ == "__main__":
    main()
