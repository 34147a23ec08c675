"""MI355X-native hot path of VIEO_SLAM (ORB extractor -> Hamming matching -> pose optimisation).

The product is libvieo_hot.so (HIP kernels + C-ABI, include/vieo_hot.h); this package only holds
the ctypes binding and the host-side mirrors of the reference's interfaces.
"""
from . import _lib  # noqa: F401
