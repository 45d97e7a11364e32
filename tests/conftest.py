import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_once():
    import __graft_entry__ as ge
    ge.build_host_and_oracle()


@pytest.fixture(scope="session", autouse=True)
def built():
    _build_once()


class Problem:
    """Mesh + DoFs + tables + parameters of one test configuration."""

    def __init__(self, kind, dim, level=1, n_sub=1, nu=1e-3, deltat=None, ordering="first_touch", **kw):
        from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables
        if kind == "cylinder":
            self.mesh = Mesh.cylinder(dim, level, **kw)
        elif kind == "cube":
            self.mesh = Mesh.cube(level)
        elif kind == "box":
            self.mesh = Mesh.box(dim, kw.get("n", [3, 3, 2][:dim]), hi=kw.get("hi", [1.0, 2.0, 1.5][:dim]))
        else:
            raise ValueError(kind)
        if n_sub > 1:
            self.mesh.partition(1, n_sub)
        self.dofs = DoFs(self.mesh, ordering)
        self.tables = Tables(dim)
        self.dim, self.nu, self.kind = dim, nu, kind
        self.deltat = deltat if deltat is not None else (2e-4 if dim == 3 else 1e-2)

    def oracle(self):
        import oracle
        return oracle.Oracle(self.dofs, self.tables, self.nu, self.deltat)

    def device(self):
        from navierstokes_project_nm4pde_amd.nsx import Nsx
        return Nsx(self.dofs, self.tables, self.nu, self.deltat)

    def smooth_velocity(self, seed=1234, amp=1.0):
        """A non-trivial divergence-carrying u_n: polynomial profile + seeded noise."""
        d = self.dofs
        X = d.support_points
        rng = np.random.default_rng(seed)
        u = np.zeros(d.n_dofs)
        if self.kind == "cube":  # Ethier-Steinmann interpolant at t = 0 (+ noise): consistent with its Dirichlet data
            from navierstokes_project_nm4pde_amd.problem import EthierSteinmann
            ex = EthierSteinmann(self.nu)
            vel = ex.velocity(X[:d.n_u])
            u[:d.n_u] = amp * vel[np.arange(d.n_u), np.arange(d.n_u) % 3] + 1e-3 * rng.standard_normal(d.n_u)
            u[d.n_u:] = ex.pressure(X[d.n_u:])
            return u
        for c in range(self.dim):
            Xc = X[c:d.n_u:self.dim]
            u[c:d.n_u:self.dim] = amp * ((c + 1) * Xc[:, 1] * (0.41 - Xc[:, 1]) + 0.3 * np.sin(3 * Xc[:, 0] + c) +
                                         1e-3 * rng.standard_normal(len(Xc)))
        u[d.n_u:] = 0.1 * rng.standard_normal(d.n_p)
        return u


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def entry_err(a, b, rowptr, floor=1e-3, gfloor=1e-4):
    """PER-ENTRY relative error of the values a against b on a CSR graph (SURVEY 8c pin 4: "every CSR value"): each entry is
    measured against max(|b_ij|, floor * largest |b| of its row, gfloor * largest |b| of the matrix), so an entry 1e-4 of the
    matrix norm that is 100 % wrong shows up as an error of order 1, where rel_err (global maximum norm) would report 1e-4.
    The global floor keeps entries that vanish in exact arithmetic from being compared with themselves: whole rows of a
    divergence block are sums of terms of the size of the largest entry that cancel to residues of 1e-19, and two correct
    summation orders differ there by an ulp of the TERMS (3e-16 of the largest entry), not of the residue."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    rowptr = np.asarray(rowptr, dtype=np.int64)
    if len(b) == 0:
        return 0.0
    counts = np.diff(rowptr)
    rowmax = np.maximum.reduceat(np.abs(b), np.minimum(rowptr[:-1], len(b) - 1))
    rowmax[counts == 0] = 0.0
    den = np.maximum(np.maximum(np.abs(b), floor * np.repeat(rowmax, counts)), gfloor * np.max(np.abs(b)))
    den[den == 0] = 1.0
    return float(np.max(np.abs(a - b) / den))


def record(name, **values):
    """Measured maxima of the parity suite (DESIGN.md section 5 quotes them): appended to gpurun_out/parity_maxima.jsonl when
    that scratch directory exists (the GPU box's copy of the repository has it; nothing is written elsewhere)."""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    if not os.path.isdir(d):
        return
    with open(os.path.join(d, "parity_maxima.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **{k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in values.items()}}) + "\n")
