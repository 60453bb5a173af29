"""``from nsynth import NsynthDataReader`` shim (teacher.py:8, student.py:9): the TensorFlow-free TFRecord reader."""
import importlib as _il
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
NsynthDataReader = _il.import_module("sr-wavenet_amd.nsynth").NsynthDataReader
