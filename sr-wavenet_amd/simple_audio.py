"""Synthetic wave batches for the reference's drivers (``simple_audio.py`` of the reference: generator.py:10,84 calls
``generate_wave_batch``).  Host-side NumPy only; same call surface and label convention, own formulas:

  Sine / Square / Sawtooth / Triangle(frequency, duration, sample_rate)   one period shapes on np.linspace(0, duration, n)
  Normalize(t, min_val, max_val)                                          affine map of [min t, max t] onto the range
  generate_wave_batch(batch_size, length, combos=False) -> (x [B, length], y [B, 10])
      per clip: a frequency f in 22..39, one of the four shapes at `length` samples per second for one second,
      N(0, 0.05) noise, normalised to [-1, 1]; label = one-hot of int(f/2 - 1) - 10   (simple_audio.py:40-61)
"""
from __future__ import annotations

import numpy as np


def CreateTicks(duration, sample_rate):
    return np.linspace(0, duration, int(sample_rate * duration))


def _phase(frequency, duration, sample_rate):
    return CreateTicks(duration, sample_rate) * frequency      # in periods


def Sine(frequency, duration, sample_rate=11025, detune=0):
    return np.sin(2.0 * np.pi * _phase(frequency, duration, sample_rate))


def Sawtooth(frequency, duration, sample_rate=11025, detune=0):
    p = _phase(frequency, duration, sample_rate)
    return 2.0 * (p - np.floor(p)) - 1.0                         # rises -1 -> 1 over each period


def Square(frequency, duration, sample_rate=11025, detune=0):
    p = _phase(frequency, duration, sample_rate)
    return np.where((p - np.floor(p)) < 0.5, 1.0, -1.0)


def Triangle(frequency, duration, sample_rate=11025, detune=0):
    f = _phase(frequency, duration, sample_rate)
    f = f - np.floor(f)
    return np.where(f < 0.5, 4.0 * f - 1.0, 3.0 - 4.0 * f)       # sawtooth with width 0.5


def Normalize(t, min_val=0, max_val=1):
    lo, hi = np.min(t), np.max(t)
    return (t - lo) / (hi - lo) * (max_val - min_val) + min_val


_FUNCS = (Sine, Square, Sawtooth, Triangle)


def generate_random_wave_f(length, combos=False, rng=None):
    rng = np.random if rng is None else rng
    frequency = int(rng.randint(18)) + 22
    labels = np.zeros(10)
    labels[int(frequency / 2 - 1) - 10] = 1
    wave = _FUNCS[int(rng.randint(len(_FUNCS)))](frequency=frequency, duration=1, sample_rate=length)
    wave = wave + rng.normal(0, 0.05, wave.shape)
    return Normalize(wave, min_val=-1, max_val=1), labels


def generate_wave_batch(batch_size, length, combos=False, rng=None):
    x, y = zip(*[generate_random_wave_f(length, combos, rng) for _ in range(batch_size)])
    return np.array(x), np.array(y)
