"""Does the layout of the conditioning biases pace the conditioned forward group kernel?  Times one 5-layer group launch of a
flow-like stack (64 channels, one conditioning frame per sample) with the biases (a) as one [rows, L*R] matrix -- a layer's
128-byte rows are 3 840 bytes apart (the layout up to r03) -- (b) layer by layer, [L][rows][R], as the engine keeps them
now, and (c) without conditioning.  Measured: 83.4 / 73.0 / 52.1 us."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
K = importlib.import_module("sr-wavenet_amd.kernels")
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
B, T = 8, 16000
pool = int(sys.argv[1]) if len(sys.argv) > 1 else 1      # 1: a bias row per time step; 125: per frame (the student)
F = T // pool
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, cond_channels=16, pool_stride=pool,
                     shift_input=True, dtype=torch.bfloat16)
os.environ["SRWN_FUSE_WT"] = "0"
e = EG.WaveNetEngine(cfg, B, T, "cuda")
e.set_inputs(torch.zeros(B, T, device="cuda"), torch.zeros(B, T, dtype=torch.int32, device="cuda"), torch.randn(B, F, 16, device="cuda"))
e.forward(); torch.cuda.synchronize()
v = e.view
def run(l0, l1, cond3, offs):
    K.residual_group_fwd(e.xs[l0], e.xs[l0 + 1:l1 + 1], e.zs[l0:l1], [e.wptr(e.o_conv[l]) for l in range(l0, l1)],
                         [e.wptr(e.o_res[l]) for l in range(l0, l1)], [v("BF")[l] for l in range(l0, l1)],
                         [v("BR")[l] for l in range(l0, l1)], e.dil[l0:l1], e.Kw, cond=cond3, cond_channel_offsets=offs,
                         pool_stride=pool, seg_rows=e.seg_rows)
for (l0, l1) in e.groups[:2]:
    wide = torch.randn(B, F, e.L * e.R, device="cuda").to(torch.bfloat16)
    cases = {"[rows, L*R]": (wide, [(l + 1) * e.R for l in range(l0, l1)]),
             "[L][rows][R] (as kept)": ([e.cond_all[l + 1].view(B, F, e.R) for l in range(l0, l1)], None),
             "no conditioning": (None, None)}
    for name, (c3, offs) in cases.items():
        for _ in range(3): run(l0, l1, c3, offs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): run(l0, l1, c3, offs)
        torch.cuda.synchronize()
        print("layers %2d..%2d  %-24s %.1f us" % (l0, l1 - 1, name, (time.perf_counter() - t0) / 20 * 1e6), flush=True)
