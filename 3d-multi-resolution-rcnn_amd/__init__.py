"""MI355X-native hot path of arthur801031/3d-multi-resolution-rcnn.

Layout:
  csrc/        hand-written HIP kernels (gfx950) + the C ABI declared in include/mr3d.h
  _hip.py      ctypes loader of libmr3d_hip.so -- raises if the library is missing (no CPU fallback)
  mmdet_ops/   host-side mirror of the reference's `mmdet.ops` plugin surface
"""
import sys as _sys

# make `import mr3d_amd.<sub>` work no matter how this package was first imported
_sys.modules.setdefault('mr3d_amd', _sys.modules[__name__])

from . import _hip  # noqa: E402,F401
from . import mmdet_ops  # noqa: E402,F401
from . import mmdet_ops as ops  # noqa: E402,F401  (`from mmdet import ops` spelling)
from . import core  # noqa: E402,F401

__version__ = '0.1.0'
