"""Alias: ``import ugrt`` == importlib.import_module("uniformgrid-raytracing_amd")."""
import importlib
import sys

_pkg = importlib.import_module("uniformgrid-raytracing_amd")
sys.modules[__name__] = _pkg
