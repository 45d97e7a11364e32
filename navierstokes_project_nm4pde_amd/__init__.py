"""MI355X-native time-step hot path of lelecaruso/NavierStokes_Project_NM4PDE.

Python here is plumbing only (ctypes bindings for tests / bench): the product is
``csrc/libnsx.so`` (HIP, C-ABI in ``include/nsx.h``) and the C++ host mirror in ``host/``.
"""
from . import frontend  # noqa: F401

__all__ = ["frontend"]
