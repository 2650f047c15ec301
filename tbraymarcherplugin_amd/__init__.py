"""tbraymarcherplugin_amd — MI355X-native raymarch + illumination hot path of TBRaymarcherPlugin.

The product is libtbrm.so (tbraymarcherplugin_amd/csrc, C-ABI in include/tbrm.h). This package only holds the
ctypes binding used by tests and bench.py, and the synthetic workload generators of SURVEY.md §8d.
"""
from . import abi  # noqa: F401

__all__ = ["abi"]
