"""MI355X-native WaveNet residual-stack hot path (drop-in for tachitachi/SR-WaveNet's ops.py/model.py path).

The directory name carries a hyphen, so import it with
``importlib.import_module("sr-wavenet_amd")`` (tests/bench do) or put ``sr-wavenet_amd/dropin`` on
``sys.path`` to get reference-named ``ops`` / ``model`` modules.
"""
from . import _lib  # noqa: F401  (does not load the .so until first use)

__all__ = ["_lib"]
