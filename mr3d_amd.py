"""Import alias: the package directory is named `3d-multi-resolution-rcnn_amd` (not a Python
identifier), so `import mr3d_amd` loads it through importlib and aliases it."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module('3d-multi-resolution-rcnn_amd')
sys.modules[__name__] = _pkg
