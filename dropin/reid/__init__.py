"""Top-level `reid` shim: makes `from reid import models`, `from reid.evaluator import
ATTEvaluator`, ... (mars_train.py:14-21) resolve to grl_amd.reid."""
import importlib
import sys

import grl_amd.reid as _impl

for _name in ('models', 'evaluator', 'train', 'loss', 'data'):
    try:
        _m = importlib.import_module('grl_amd.reid.' + _name)
    except ImportError:          # sub-package not provided (yet)
        continue
    sys.modules['reid.' + _name] = _m
    globals()[_name] = _m

# sub-modules callers import by path: ONE module object under both names, so that classes
# (e.g. reid.data.jpeg.JpegBatch) are the ones grl_amd itself checks with isinstance
for _sub in ('data.augment', 'data.jpeg'):
    try:
        sys.modules['reid.' + _sub] = importlib.import_module('grl_amd.reid.' + _sub)
    except ImportError:
        pass
