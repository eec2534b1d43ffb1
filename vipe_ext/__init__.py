"""`vipe_ext` - the module name the reference's loader binds (vipe/ext/__init__.py:24-46: `import vipe_ext as _C`, then
`_C.droid_net_ext`, `_C.grounding_dino_ext`, `_C.utils_ext`, `_C.slam_ext`, `_C.scatter_ext`, `_C.lietorch_ext`,
`_C.corr_ext`; the pybind11 module of csrc/bind.cpp:28-49).  With this repository on PYTHONPATH the unmodified loader
finds this package instead of JIT-compiling csrc/: the seven attributes are the MI355X operator modules over the C ABI
of libvipe_amd.so (include/vipe_amd.h).  `grounding_dino_ext` exists and raises (outside the path)."""

from vipe_amd.ext import (corr_ext, droid_net_ext, grounding_dino_ext, lietorch_ext, scatter_ext, slam_ext,  # noqa: F401
                          utils_ext)

__all__ = ["droid_net_ext", "grounding_dino_ext", "utils_ext", "slam_ext", "scatter_ext", "lietorch_ext", "corr_ext"]
