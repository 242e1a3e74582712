"""nano-kazen_amd — MI355X-native path-tracing core for nano-kazen's path_mis hot path.

Python here is harness plumbing around the C ABI in include/kazen_mi355x.h (ctypes): the
product is csrc/ (hand-written HIP for gfx950 + host BVH builder + the C-ABI shim).
Import with importlib.import_module("nano-kazen_amd") (the directory name has a hyphen).
"""
from . import abi, output, scenes, shard, xmlscene     # noqa: F401
from .render import Scene            # noqa: F401
