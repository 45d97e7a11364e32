"""Problem data of the three reference executables, as plain numpy formulas.

Mirrors the `Function<dim>` functors of the reference headers (host-side scalar formulas):
  * InletVelocity  — reference include/NavierStokes3D.hpp:17-81, include/NavierStokes2D.hpp:18-81
  * ExactSolution / FunctionH (Ethier-Steinmann) — reference include/Convergence3D.hpp:51-201
and the boundary-value map built in NavierStokes::assemble / assemble_time_step
(reference src/NavierStokes3D.cpp:327-352, 515-540; src/Convergence3D.cpp:359-378).
"""
import numpy as np


class InletVelocity:
    """Parabolic inflow; `u_m` is a hard-coded member in the reference (9.0 in 3D, 1.5 in 2D) and a parameter here."""

    def __init__(self, dim, test_case=2, u_m=None, H=0.41):
        self.dim, self.test_case, self.H = dim, test_case, H
        self.u_m = (9.0 if dim == 3 else 1.5) if u_m is None else u_m
        self.time = 0.0

    def set_time(self, t):
        self.time = t

    def value(self, p):
        """x-component at points p[n, dim]; the other components are zero."""
        H, um = self.H, self.u_m
        if self.test_case == 1:
            return np.zeros(len(p))
        if self.dim == 3:
            v = 16.0 * um * p[:, 1] * p[:, 2] * (H - p[:, 2]) * (H - p[:, 1]) / (H * H * H * H)
            if self.test_case == 3:   # NavierStokes3D.hpp:32-34
                v = 16.0 * um * p[:, 1] * p[:, 2] * (H - p[:, 2]) * (H - p[:, 1]) * np.sin(np.pi * self.time / 8.0) / (H * H * H * H)
            return v
        if self.test_case == 2:       # NavierStokes2D.hpp:33-34 (case 2 is the sinusoidal one in 2D)
            return 4.0 * um * p[:, 1] * (H - p[:, 1]) * np.sin(np.pi * self.time / 8.0) / (H * H)
        return 4.0 * um * p[:, 1] * (H - p[:, 1]) / (H * H)

    def mean_velocity(self):
        """getMeanVelocity(): NavierStokes3D.hpp:64-75 / NavierStokes2D.hpp:64-76 (case 2 is constant in BOTH)."""
        if self.test_case == 1:
            return 0.0
        k = 4.0 / 9.0 if self.dim == 3 else 2.0 / 3.0
        if self.test_case == 3:
            return k * self.u_m * np.sin(self.time * np.pi / 8.0)
        return k * self.u_m


class EthierSteinmann:
    """Exact solution of the convergence executable (Convergence3D.hpp:51-148), a=pi/4, b=pi/2, nu=1e-2."""

    def __init__(self, nu=1e-2):
        self.nu, self.a, self.b, self.time = nu, np.pi / 4.0, np.pi / 2.0, 0.0

    def set_time(self, t):
        self.time = t

    def velocity(self, p):
        a, b, e = self.a, self.b, np.exp(-self.nu * self.b * self.b * self.time)
        x, y, z = p[:, 0], p[:, 1], p[:, 2]
        u = np.empty((len(p), 3))
        u[:, 0] = -a * e * (np.exp(a * x) * np.sin(a * y + b * z) + np.exp(a * z) * np.cos(a * x + b * y))
        u[:, 1] = -a * e * (np.exp(a * y) * np.sin(a * z + b * x) + np.exp(a * x) * np.cos(a * y + b * z))
        u[:, 2] = -a * e * (np.exp(a * z) * np.sin(a * x + b * y) + np.exp(a * y) * np.cos(a * z + b * x))
        return u

    def pressure(self, p):
        a, b = self.a, self.b
        x, y, z = p[:, 0], p[:, 1], p[:, 2]
        factor = -(a * a * np.exp(-2 * self.nu * b * b * self.time)) / 2.0
        t1 = 2.0 * np.sin(a * x + b * y) * np.cos(a * z + b * x) * np.exp(a * (y + z))
        t2 = 2.0 * np.sin(a * y + b * z) * np.cos(a * x + b * y) * np.exp(a * (x + z))
        t3 = 2.0 * np.sin(a * z + b * x) * np.cos(a * y + b * z) * np.exp(a * (x + y))
        t4 = np.exp(2 * a * x) + np.exp(2 * a * y) + np.exp(2 * a * z)
        return factor * (t1 + t2 + t3 + t4)

    def gradient(self, p):
        """grad[n, i, j] = d u_i / d x_j  (Convergence3D.hpp:109-132)."""
        a, b, e = self.a, self.b, np.exp(-self.nu * self.b * self.b * self.time)
        x, y, z = p[:, 0], p[:, 1], p[:, 2]
        g = np.empty((len(p), 3, 3))
        ex, ey, ez = np.exp(a * x), np.exp(a * y), np.exp(a * z)
        g[:, 0, 0] = -a * e * (a * ex * np.sin(a * y + b * z) - a * ez * np.sin(a * x + b * y))
        g[:, 0, 1] = -a * e * (a * ex * np.cos(a * y + b * z) - b * ez * np.sin(a * x + b * y))
        g[:, 0, 2] = -a * e * (b * ex * np.cos(a * y + b * z) + a * ez * np.cos(a * x + b * y))
        g[:, 1, 0] = -a * e * (b * ey * np.cos(a * z + b * x) + a * ex * np.cos(a * y + b * z))
        g[:, 1, 1] = -a * e * (a * ey * np.sin(a * z + b * x) - a * ex * np.sin(a * y + b * z))
        g[:, 1, 2] = -a * e * (a * ey * np.cos(a * z + b * x) - b * ex * np.sin(a * y + b * z))
        g[:, 2, 0] = -a * e * (a * ez * np.cos(a * x + b * y) - b * ey * np.sin(a * z + b * x))
        g[:, 2, 1] = -a * e * (b * ez * np.cos(a * x + b * y) + a * ey * np.cos(a * z + b * x))
        g[:, 2, 2] = -a * e * (a * ez * np.sin(a * x + b * y) - a * ey * np.sin(a * z + b * x))
        return g

    def neumann_h(self, p):
        """FunctionH on y = -1 (Convergence3D.hpp:159-174): nu * du/dn - p n with n = (0,-1,0) folded into the signs."""
        a, b, nu, e = self.a, self.b, self.nu, np.exp(-self.nu * self.b * self.b * self.time)
        x, y, z = p[:, 0], p[:, 1], p[:, 2]
        h = np.empty((len(p), 3))
        h[:, 0] = -nu * a * e * (a * np.exp(a * x) * np.cos(a * y + b * z) - b * np.exp(a * z) * np.sin(a * x + b * y))
        h[:, 1] = -nu * a * e * (a * np.exp(a * y) * np.sin(a * z + b * x) - a * np.exp(a * x) * np.sin(a * y + b * z)) - self.pressure(p)
        h[:, 2] = -nu * a * e * (b * np.exp(a * z) * np.cos(a * x + b * y) + a * np.exp(a * y) * np.cos(a * z + b * x))
        return h


def cylinder_boundary_values(dofs, inlet, time):
    """boundary_values map of the cylinder executables, as sorted (dof, value) arrays.

    First the inlet (id 0) with the time-dependent profile, then walls + obstacle (ids 2, 3) with zero, the second
    call overwriting shared dofs exactly as std::map assignment does (NavierStokes3D.cpp:331-351, 519-539)."""
    inlet.set_time(time)
    dim = dofs.dim
    bv = {}
    d0 = dofs.boundary_dofs(0)
    vals = np.zeros(len(d0))
    xcomp = (d0 % dim) == 0
    vals[xcomp] = inlet.value(dofs.support_points[d0[xcomp]])
    bv.update(zip(d0.tolist(), vals.tolist()))
    for bid in (2, 3):
        for d in dofs.boundary_dofs(bid).tolist():
            bv[d] = 0.0
    keys = np.array(sorted(bv), dtype=np.int32)
    return keys, np.array([bv[k] for k in keys.tolist()], dtype=np.float64)


def ethier_boundary_values(dofs, exact, time):
    """Dirichlet ids 0,1,2,4,5 with the exact solution (Convergence3D.cpp:363-378); id 3 is Neumann."""
    exact.set_time(time)
    ds = np.unique(np.concatenate([dofs.boundary_dofs(b) for b in (0, 1, 2, 4, 5)])).astype(np.int32)
    u = exact.velocity(dofs.support_points[ds])
    return ds, u[np.arange(len(ds)), ds % 3].astype(np.float64)
