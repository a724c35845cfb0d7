"""Top-level `utils` shim (mars_train.py:14-17 imports utils.logging / utils.serialization)."""
import importlib
import sys

from grl_amd.utils import to_numpy, to_torch  # noqa: F401

for _name in ('logging', 'meters', 'osutils', 'serialization'):
    _m = importlib.import_module('grl_amd.utils.' + _name)
    sys.modules['utils.' + _name] = _m
    globals()[_name] = _m
