"""ctypes loaders for the in-tree shared libraries."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)

HOST_SO = os.path.join(_HERE, "host", "libnsx_host.so")
DEV_SO = os.path.join(_HERE, "csrc", "libnsx.so")

_cache = {}


def load(path):
    if path not in _cache:
        if not os.path.exists(path):
            raise OSError(
                "%s is missing: run `make` (or `python -c 'import __graft_entry__ as g; g.build()'`) "
                "at the repository root" % path)
        _cache[path] = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    return _cache[path]
