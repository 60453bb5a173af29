"""Where the skip-sum GEMM's time goes: the same launch (a) as it runs, (b) without the gate recomputation (PRO_NONE), (c) with
every layer's chunk reading layer 0's rows (activations from L2 / Infinity Cache instead of HBM), (d) both."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
K = importlib.import_module("sr-wavenet_amd.kernels")
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
e = EG.WaveNetEngine(cfg, 8, 16000, "cuda")
a = torch.randn(8, 16000, device="cuda").clamp(-1, 1) * 0.5
e.set_inputs(a, torch.randint(0, 256, (8, 16000), dtype=torch.int32, device="cuda"))
e.forward(); torch.cuda.synchronize()
N, R, L, S = e.N, e.R, e.L, e.S
def run(pro, cstride):
    K.pw_linear(e.zs.data_ptr(), R, cstride, R, L * R, e.wptr(e.o_skip), e.bs_sum, e.r0, S, S, N, pro=pro, epi=K.EPI_RELU)
for name, pro, cs in (("as it runs", K.PRO_GATE, N * R), ("no gate", K.PRO_NONE, N * R), ("one layer's rows", K.PRO_GATE, 0),
                      ("no gate, one layer's rows", K.PRO_NONE, 0)):
    for _ in range(3): run(pro, cs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): run(pro, cs)
    torch.cuda.synchronize()
    print("%-28s %.1f us" % (name, (time.perf_counter() - t0) / 20 * 1e6), flush=True)
