"""``from simple_audio import generate_wave_batch`` shim (generator.py:10)."""
import importlib as _il
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
_m = _il.import_module("sr-wavenet_amd.simple_audio")
generate_wave_batch = _m.generate_wave_batch
generate_random_wave_f = _m.generate_random_wave_f
Sine, Square, Sawtooth, Triangle, Normalize, CreateTicks = _m.Sine, _m.Square, _m.Sawtooth, _m.Triangle, _m.Normalize, _m.CreateTicks
