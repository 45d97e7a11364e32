"""CPU suite: the C-ABI shared library loads and exports every symbol include/nsx.h declares; no CPU fallback exists."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libnsx():
    import __graft_entry__ as ge
    ge.build()                       # hipcc cross-compiles gfx950 without a GPU
    from navierstokes_project_nm4pde_amd._lib import DEV_SO
    return ctypes.CDLL(DEV_SO)


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nsxh?_[a-z0-9_]+)\s*\(", txt)))


def test_device_library_exports_every_declared_symbol(libnsx):
    from navierstokes_project_nm4pde_amd import nsx
    names = _declared("nsx.h")
    assert len(names) >= 30
    assert sorted(nsx.API) == names
    for n in names:
        assert hasattr(libnsx, n), n


def test_host_library_exports_every_declared_symbol():
    from navierstokes_project_nm4pde_amd._lib import HOST_SO
    lib = ctypes.CDLL(HOST_SO)
    for n in _declared("nsx_host.h"):
        assert hasattr(lib, n), n


def test_kernels_are_compiled_for_gfx950(libnsx):
    from navierstokes_project_nm4pde_amd._lib import DEV_SO
    blob = open(DEV_SO, "rb").read()
    assert b"gfx950" in blob and b"k_ilu_solve_lanes" in blob and b"k_cell_convection" in blob


def test_no_cpu_fallback_without_a_gpu(libnsx):
    """On a machine without a HIP device nsx_create must fail loudly (NSX_ERR_HIP), never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from navierstokes_project_nm4pde_amd import nsx
    L = nsx.lib()
    h = ctypes.c_void_p()
    prm = nsx.Params(3, 0, 1e-3, 2e-4)
    rc = L.nsx_create(ctypes.byref(prm), ctypes.byref(h))
    assert rc == -2 and not h.value
    assert b"HIP" in L.nsx_last_error(None) or b"hip" in L.nsx_last_error(None)
    bad = nsx.Params(4, 0, 1e-3, 2e-4)
    assert L.nsx_create(ctypes.byref(bad), ctypes.byref(h)) == -1


def test_product_does_not_reference_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "navierstokes_project_nm4pde_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "nsx_oracle" not in txt and "import oracle" not in txt and "liboracle" not in txt, os.path.join(dp, f)
    for hdr in ("nsx.h", "nsx_host.h"):
        assert "oracle" not in open(os.path.join(ROOT, "include", hdr)).read().lower()
