"""pyprogressivex — drop-in replacement of the reference's pybind11 module
(/root/reference/src/pyprogressivex/src/bindings.cpp:394-494) on top of libpgx.so (HIP, gfx950).

The five entry points keep the reference's names, argument order, defaults, return layout and error messages.
"""
from ._api import (find6DPoses, findFundamentalMatrices, findHomographies, findLines, findTwoViewMotions,
                   findVanishingPoints)

__all__ = ["find6DPoses", "findHomographies", "findTwoViewMotions", "findFundamentalMatrices", "findLines",
           "findVanishingPoints"]
