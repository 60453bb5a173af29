"""``from model import WaveNet`` shim: put this directory on sys.path and the
reference drivers' imports resolve to the MI355X implementation."""
import importlib as _il
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
_m = _il.import_module("sr-wavenet_amd.model")
WaveNet = _m.WaveNet
WaveNetTeacher = _m.WaveNetTeacher
WaveNetAutoEncoder = _m.WaveNetAutoEncoder
ParallelWaveNet = _m.ParallelWaveNet
