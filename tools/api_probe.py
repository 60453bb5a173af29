#!/usr/bin/env python3
"""Step time through the reference-style Python API (NumPy in, loss out) at teacher.py's / student.py's own shapes."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
M = importlib.import_module("sr-wavenet_amd.model")
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
B, T, lat, pool = 4, 4096, 16, 512
rng = np.random.default_rng(0)
x = (0.5 * np.sin(2 * np.pi * 220 * np.arange(T)[None] / 4000) + 0.05 * rng.standard_normal((B, T))).clip(-1, 1).astype(np.float32)
teacher = M.WaveNetAutoEncoder(input_size=T, condition_size=0, num_mixtures=5, dilations=dil, latent_channels=lat,
                               skip_channels=128, pool_stride=pool, learning_rate=1e-4)          # teacher.py:61
for name, n in (("first steps (eager + capture)", 3), ("steady state", 50)):
    t0 = time.perf_counter()
    for _ in range(n):
        loss = teacher.train(x)
    print("WaveNetAutoEncoder.train  %-30s %.2f ms/step  loss %.1f" % (name, (time.perf_counter() - t0) / n * 1e3, loss), flush=True)
enc = teacher.encode(x)
student = M.ParallelWaveNet(input_size=T, condition_size=0, dilations=dil, teacher=teacher, dilation_channels=32,
                            skip_channels=128, num_flows=4, latent_channels=lat, pool_stride=pool, gamma=1e-3)  # student.py:82
noise = rng.logistic(0, 1, (B, T)).astype(np.float32)
for name, n in (("first steps (eager + capture)", 3), ("steady state", 50)):
    t0 = time.perf_counter()
    for _ in range(n):
        l, p = student.train_fast(None, noise, x, enc)
    print("ParallelWaveNet.train_fast %-30s %.2f ms/step  loss %.1f" % (name, (time.perf_counter() - t0) / n * 1e3, l), flush=True)
