"""ctypes binding of the CPU oracle (oracle/nsx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package never imports this module.  PARITY UNPINNED — see oracle/nsx_oracle.h.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "liboracle.so")        # serial, reference-shaped: what the parity tests check against
SO_MT = os.path.join(_HERE, "liboracle_mt.so")  # same source with -fopenmp: bench.py's all-host-cores CPU baseline

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)

TEMAM, DOUBLE_CONVECTION = 1, 2
YOSIDA, SIMPLE, AYOSIDA, ASIMPLE = 0, 1, 2, 3


class Stats(C.Structure):
    _fields_ = [("outer_iterations", C.c_int), ("inner_F_iterations", C.c_int), ("inner_S_iterations", C.c_int),
                ("n_F_solves", C.c_int), ("n_S_solves", C.c_int), ("final_residual", C.c_double),
                ("t_prec", C.c_double), ("t_solve", C.c_double), ("status", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_libs = {}


def use_mt_library(path):
    """bench.py's cpu_baseline leg: take the OpenMP build from `path` (the same source compiled on the benchmark host with
    -O3 -march=native) instead of the portable in-tree build.  The parity tests never call this."""
    global SO_MT
    SO_MT = path


def usable_cores():
    """CPU cores this process may actually use: scheduler affinity capped by the cgroup CPU quota (a GPU box hands a
    one-GPU job a share of its host cores, not all of them)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def lib(mt=False):
    so = SO_MT if mt else SO  # looked up at call time: use_mt_library may have redirected SO_MT
    if so not in _libs:
        if not os.path.exists(so):
            raise OSError("%s missing: run `make oracle`" % so)
        L = C.CDLL(so)
        L.orc_threads.restype = C.c_int
        L.orc_set_threads.argtypes = [C.c_int]
        vp = C.c_void_p
        L.orc_create.restype = vp
        L.orc_create.argtypes = [C.c_int] * 5 + [_i32p, _f64p] + [C.c_int] * 3 + [_f64p] * 4 + \
            [C.POINTER(_i32p), C.POINTER(_i32p), C.c_double, C.c_double]
        L.orc_destroy.argtypes = [vp]
        L.orc_set_ranks.argtypes = [vp, C.c_int, _i32p, _i32p]
        L.orc_set_schur_blocks.argtypes = [vp, C.c_int, _i32p]
        L.orc_set_compact.argtypes = [vp, C.c_int]
        L.orc_assemble.argtypes = [vp, C.c_int]
        L.orc_assemble_time_step.argtypes = [vp, C.c_int]
        L.orc_add_rhs.argtypes = [vp, C.c_int, _i32p, _f64p]
        L.orc_apply_boundary_values.argtypes = [vp, C.c_int, _i32p, _f64p]
        L.orc_solve_time_step.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(Stats)]
        for f in ("orc_solution", "orc_solution_owned", "orc_rhs"):
            getattr(L, f).restype = _f64p
            getattr(L, f).argtypes = [vp]
        L.orc_matrix_values.restype = _f64p
        L.orc_matrix_values.argtypes = [vp, C.c_int, C.c_int]
        L.orc_schur.restype = C.c_int
        L.orc_schur.argtypes = [vp, C.POINTER(_i32p), C.POINTER(_i32p), C.POINTER(_f64p)]
        L.orc_ilu_F.restype = _f64p
        L.orc_ilu_F.argtypes = [vp]
        L.orc_ilu_S.restype = _f64p
        L.orc_ilu_S.argtypes = [vp]
        L.orc_compute_forces.argtypes = [vp, C.c_int, _i32p, _i32p, C.c_int, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p]
        L.orc_spmv.argtypes = [C.c_int, _i32p, _i32p, _f64p, _f64p, _f64p]
        L.orc_system_vmult.argtypes = [vp, _f64p, _f64p]
        L.orc_ilu0_factor.argtypes = [C.c_int, _i32p, _i32p, _f64p, C.c_int, _i32p, _f64p]
        L.orc_ilu0_solve.argtypes = [C.c_int, _i32p, _i32p, _f64p, C.c_int, _i32p, _f64p, _f64p]
        L.orc_prec_initialize.argtypes = [vp, C.c_int]
        L.orc_prec_vmult.argtypes = [vp, C.c_int, C.c_double, C.c_int, _f64p, _f64p, C.POINTER(Stats)]
        _libs[so] = L
    return _libs[so]


def _i(a):
    return a.ctypes.data_as(_i32p)


def _d(a):
    return a.ctypes.data_as(_f64p)


def _ci(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _cd(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    """One `NavierStokes` problem instance of the restated reference algorithm (raw arrays in, raw arrays out)."""

    def __init__(self, dofs, tables, nu, deltat, threads=1, compact=False):
        """threads = 1: the serial restatement (liboracle.so).  threads > 1: the OpenMP build of the same source
        (liboracle_mt.so) on that many threads — only bench.py's cpu_baseline leg uses it."""
        L = lib(mt=threads > 1)
        if threads > 1:
            L.orc_set_threads(int(threads))
        self.threads = L.orc_threads()
        self.L = L
        self.dim, self.n_u, self.n_p = dofs.dim, dofs.n_u, dofs.n_p
        self.n = self.n_u + self.n_p
        self.graphs = [dofs.reference_sparsity(b) for b in range(4)]
        rps = (_i32p * 4)(*[_i(g[0]) for g in self.graphs])
        cis = (_i32p * 4)(*[_i(g[1]) for g in self.graphs])
        cd, cc = _ci(dofs.cell_dofs), _cd(dofs.cell_coords)
        N2, dN2, N1, w = _cd(tables.N2), _cd(tables.dN2), _cd(tables.N1), _cd(tables.weights)
        self._h = L.orc_create(dofs.dim, dofs.n_cells, dofs.dofs_per_cell, dofs.n_u, dofs.n_p, _i(cd), _d(cc),
                               tables.n_q, tables.n_p2, tables.n_p1, _d(N2), _d(dN2), _d(N1), _d(w), rps, cis,
                               float(nu), float(deltat))
        if dofs.n_subdomains > 1:
            self.set_ranks(dofs.owned_u_ptr, dofs.owned_p_ptr)
        if compact:  # scalar P2 operator for the products with system(0,0) and its ILU(0): bench.py's "best CPU" leg, not the reference's layout
            L.orc_set_compact(self._h, 1)

    def __del__(self):
        try:
            self.L.orc_destroy(self._h)
        except Exception:
            pass

    def set_ranks(self, u_ptr, p_ptr):
        u_ptr, p_ptr = _ci(u_ptr), _ci(p_ptr)
        self.L.orc_set_ranks(self._h, len(u_ptr) - 1, _i(u_ptr), _i(p_ptr))

    def set_schur_blocks(self, p_ptr):
        p_ptr = _ci(p_ptr)
        self.L.orc_set_schur_blocks(self._h, len(p_ptr) - 1, _i(p_ptr))

    # -- state ---------------------------------------------------------------------------------
    def _vec(self, fn):
        return np.ctypeslib.as_array(fn(self._h), shape=(self.n,))

    @property
    def solution(self):
        return self._vec(self.L.orc_solution)

    @property
    def solution_owned(self):
        return self._vec(self.L.orc_solution_owned)

    @property
    def rhs(self):
        return self._vec(self.L.orc_rhs)

    def matrix(self, which, block):
        """values of `which` (0 system, 1 mass, 2 convection, 3 stiffness, 4 pressure mass) in block's CSR graph."""
        g = self.graphs[3 if which == 4 else block]
        return np.ctypeslib.as_array(self.L.orc_matrix_values(self._h, which, block), shape=(len(g[1]),))

    def scipy(self, which, block):
        import scipy.sparse as sp
        g = self.graphs[3 if which == 4 else block]
        ncols = [self.n_u, self.n_p, self.n_u, self.n_p][3 if which == 4 else block]
        return sp.csr_matrix((self.matrix(which, block).copy(), g[1], g[0]), shape=(len(g[0]) - 1, ncols))

    # -- the reference's member functions --------------------------------------------------------
    def assemble(self, flags=0):
        self.L.orc_assemble(self._h, flags)

    def assemble_time_step(self, flags=0):
        self.L.orc_assemble_time_step(self._h, flags)

    def add_rhs(self, dofs, vals):
        dofs, vals = _ci(dofs), _cd(vals)
        self.L.orc_add_rhs(self._h, len(dofs), _i(dofs), _d(vals))

    def apply_boundary_values(self, dofs, vals):
        dofs, vals = _ci(dofs), _cd(vals)
        self.L.orc_apply_boundary_values(self._h, len(dofs), _i(dofs), _d(vals))

    def solve_time_step(self, prec=YOSIDA, tol_abs=1e-4, inner_rtol=1e-2, maxiter=100000, inner_maxiter=100000):
        st = Stats()
        self.L.orc_solve_time_step(self._h, prec, tol_abs, inner_rtol, maxiter, inner_maxiter, C.byref(st))
        return st.as_dict()

    def prec_initialize(self, prec):
        self.L.orc_prec_initialize(self._h, prec)

    def prec_vmult(self, prec, src, inner_rtol=1e-2, inner_maxiter=100000):
        src = _cd(src)
        dst = np.zeros_like(src)
        st = Stats()
        self.L.orc_prec_vmult(self._h, prec, inner_rtol, inner_maxiter, _d(dst), _d(src), C.byref(st))
        return dst, st.as_dict()

    def compute_forces(self, cells, lfaces, ftab):
        cells, lfaces = _ci(cells), _ci(lfaces)
        N2, dN2, N1, w = _cd(ftab.N2), _cd(ftab.dN2), _cd(ftab.N1), _cd(ftab.weights[:ftab.n_qf])
        d, l = C.c_double(), C.c_double()
        self.L.orc_compute_forces(self._h, len(cells), _i(cells), _i(lfaces), ftab.n_qf, _d(N2), _d(dN2), _d(N1), _d(w),
                                  C.byref(d), C.byref(l))
        return d.value, l.value

    def system_vmult(self, src):
        src = _cd(src)
        dst = np.zeros_like(src)
        self.L.orc_system_vmult(self._h, _d(dst), _d(src))
        return dst

    def schur(self):
        import scipy.sparse as sp
        rp, ci, v = _i32p(), _i32p(), _f64p()
        n = self.L.orc_schur(self._h, C.byref(rp), C.byref(ci), C.byref(v))
        rowptr = np.ctypeslib.as_array(rp, shape=(n + 1,)).copy()
        nnz = int(rowptr[-1])
        return sp.csr_matrix((np.ctypeslib.as_array(v, shape=(nnz,)).copy(),
                              np.ctypeslib.as_array(ci, shape=(nnz,)).copy(), rowptr), shape=(n, n))

    def ilu_F(self):
        return np.ctypeslib.as_array(self.L.orc_ilu_F(self._h), shape=(len(self.graphs[0][1]),)).copy()

    def ilu_S(self, nnz):
        return np.ctypeslib.as_array(self.L.orc_ilu_S(self._h), shape=(nnz,)).copy()


def spmv(rowptr, colind, vals, x):
    rowptr, colind, vals, x = _ci(rowptr), _ci(colind), _cd(vals), _cd(x)
    y = np.zeros(len(rowptr) - 1)
    lib().orc_spmv(len(rowptr) - 1, _i(rowptr), _i(colind), _d(vals), _d(x), _d(y))
    return y


def ilu0_factor(rowptr, colind, vals, block_ptr):
    rowptr, colind, vals, block_ptr = _ci(rowptr), _ci(colind), _cd(vals), _ci(block_ptr)
    out = np.zeros_like(vals)
    lib().orc_ilu0_factor(len(rowptr) - 1, _i(rowptr), _i(colind), _d(vals), len(block_ptr) - 1, _i(block_ptr), _d(out))
    return out


def ilu0_solve(rowptr, colind, lu, block_ptr, b):
    rowptr, colind, lu, block_ptr, b = _ci(rowptr), _ci(colind), _cd(lu), _ci(block_ptr), _cd(b)
    x = np.zeros_like(b)
    lib().orc_ilu0_solve(len(rowptr) - 1, _i(rowptr), _i(colind), _d(lu), len(block_ptr) - 1, _i(block_ptr), _d(b), _d(x))
    return x
