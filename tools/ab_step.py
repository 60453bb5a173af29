#!/usr/bin/env python3
"""A/B of library builds on ONE GPU box (box-to-box spread exceeds most single changes): for every library given, the
config-2 training step replayed as a hipGraph (ms/step, best and median of `--rounds` alternating rounds) and the
per-kernel spans of a few eager steps.  Each measurement is a fresh child process (SRWN_LIB_PATH is read at import).

  python tools/ab_step.py base=sr-wavenet_amd/libsrwn.so prio=ab/libsrwn_prio.so [--rounds 3] [--steps 200]
  a variant may carry environment settings: nodefer=sr-wavenet_amd/libsrwn.so,SRWN_DEFER_LOSS=0
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import importlib, json, os, sys, time
sys.path.insert(0, %(root)r)
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
res = []
for rep in range(%(reps)d):
    t0 = time.perf_counter()
    for _ in range(%(steps)d): eng.train_step_graphed()
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / %(steps)d * 1e3)
eng.timing = True; eng.spans.clear()
for _ in range(5): eng.train_step()
torch.cuda.synchronize()
spans = {k: float(np.median([sum(s.elapsed_time(e) for s, e in v[i * (len(v) // 5):(i + 1) * (len(v) // 5)]) for i in range(5)])) for k, v in eng.spans.items()}
print("ABRESULT " + json.dumps({"ms": res, "spans": spans, "loss": float(eng.loss.item())}))
"""


def main():
    args = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 200
    libs = [a.split("=", 1) for a in args]
    out = {n: {"ms": [], "spans": []} for n, _ in libs}
    for r in range(rounds):
        for name, spec in libs:
            env = dict(os.environ)
            path, *sets = spec.split(",")
            for kv in sets:
                k, v = kv.split("=", 1)
                env[k] = v
            p = os.path.join(ROOT, path)
            if os.path.abspath(p) != os.path.join(ROOT, "sr-wavenet_amd", "libsrwn.so"):
                env["SRWN_LIB_PATH"] = p
            else:
                env.pop("SRWN_LIB_PATH", None)
            pr = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "reps": 3, "steps": steps}], env=env, cwd=ROOT,
                                capture_output=True, text=True, timeout=600)
            line = [l for l in pr.stdout.splitlines() if l.startswith("ABRESULT ")]
            if pr.returncode or not line:
                print("FAILED", name, pr.stderr[-2000:], flush=True)
                continue
            d = json.loads(line[0][9:])
            out[name]["ms"] += d["ms"]
            out[name]["spans"].append(d["spans"])
            print("round %d %-10s ms/step %s  loss %.6f" % (r, name, " ".join("%.4f" % m for m in d["ms"]), d["loss"]), flush=True)
    import numpy as np
    print()
    keys = sorted({k for n in out for s in out[n]["spans"] for k in s})
    print("%-12s %9s %9s  " % ("variant", "best ms", "median") + " ".join("%14s" % k[:14] for k in keys))
    for n, _ in libs:
        if not out[n]["ms"]:
            continue
        sp = {k: np.median([s[k] for s in out[n]["spans"] if k in s]) for k in keys}
        print("%-12s %9.4f %9.4f  " % (n, min(out[n]["ms"]), float(np.median(out[n]["ms"]))) + " ".join("%14.4f" % sp[k] for k in keys))


if __name__ == "__main__":
    main()
