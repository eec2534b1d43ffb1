"""Drop-in for the reference's `vipe.ext` package (vipe/ext/__init__.py:24-46).

The reference imports one pybind11 module `vipe_ext` and re-exports its seven submodules; callers do
`from vipe.ext import droid_net_ext, slam_ext, lietorch_ext, scatter_ext, corr_ext`.  Here each
submodule is a thin Python module over the C ABI of libvipe_amd.so (include/vipe_amd.h): same
function names, argument order, return structure and error behaviour.
"""

from . import corr_ext, droid_net_ext, lietorch_ext, scatter_ext, slam_ext  # noqa: F401
from ._out_of_scope import grounding_dino_ext  # noqa: F401
from . import utils_ext  # noqa: F401
