"""Imports the hyphen-named product package for the tests."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = importlib.import_module("sr-wavenet_amd")


def sub(name):
    return importlib.import_module("sr-wavenet_amd." + name)
