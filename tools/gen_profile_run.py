import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
EG = importlib.import_module("sr-wavenet_amd.engine")
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=256, shift_input=True, dtype=torch.bfloat16)
eng = EG.WaveNetEngine(cfg, 1, 64, "cuda")
for B in (32, 2048):
    eng.generate(4000, mode="sample", seed=1, batch=B)
pool, lat, Mx = 125, 16, 10
cfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=4 * Mx, cond_channels=lat,
                     pool_stride=pool, shift_input=True, head_mode="mol", dtype=torch.bfloat16)
eng = EG.WaveNetEngine(cfg, 1, pool, "cuda")
for B in (32, 2048):
    eng.generate(4000, mode="sample", seed=1, batch=B, cond=torch.randn((B, 4000 // pool, lat), device="cuda"))
torch.cuda.synchronize()
