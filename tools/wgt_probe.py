"""Time of srwn_wgrad_skip_wt alone on the benchmark's shapes (after one forward + backward filled its inputs)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
K = importlib.import_module("sr-wavenet_amd.kernels")
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
eng = EG.WaveNetEngine(cfg, 8, 16000, "cuda")
a = torch.randn(8, 16000, device="cuda").clamp(-1, 1) * 0.5
eng.set_inputs(a, torch.randint(0, 256, (8, 16000), dtype=torch.int32, device="cuda"))
eng.forward(); eng.backward()
torch.cuda.synchronize()
def run():
    K.wgrad_skip_wt(eng.cTs, eng.wt_layer_st, eng.wt_layer_seg, eng.dtotal, eng.wg_parts, eng.wg_bparts, eng.ns_skip_wt, eng.B, eng.T, eng.R)
for dbg in (sys.argv[1:] or ["0"]):
    os.environ["SRWN_WGT_DBG"] = dbg
    for _ in range(3): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize()
    print("dbg", dbg, "%.1f us" % ((time.perf_counter() - t0) / 20 * 1e6), flush=True)
