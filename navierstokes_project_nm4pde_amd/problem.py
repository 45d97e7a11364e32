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
    cache = getattr(dofs, "_cylinder_bc", None)
    if cache is None:  # the dof set and who wins on shared dofs do not change with time: only the inlet values do
        d0 = dofs.boundary_dofs(0)
        zero = np.unique(np.concatenate([dofs.boundary_dofs(2), dofs.boundary_dofs(3)]))
        keys = np.union1d(d0, zero).astype(np.int32)
        inlet_x = np.isin(keys, d0) & ~np.isin(keys, zero) & (keys % dim == 0)
        cache = dofs._cylinder_bc = (keys, inlet_x, np.ascontiguousarray(dofs.support_points[keys[inlet_x]]))
    keys, inlet_x, pts = cache
    vals = np.zeros(len(keys))
    vals[inlet_x] = inlet.value(pts)
    return keys, vals


def ethier_boundary_values(dofs, exact, time):
    """Dirichlet ids 0,1,2,4,5 with the exact solution (Convergence3D.cpp:363-378); id 3 is Neumann."""
    exact.set_time(time)
    ds = np.unique(np.concatenate([dofs.boundary_dofs(b) for b in (0, 1, 2, 4, 5)])).astype(np.int32)
    u = exact.velocity(dofs.support_points[ds])
    return ds, u[np.arange(len(ds)), ds % 3].astype(np.float64)


# ------------------------------------------------------------------ host-side integrals of the convergence executable
_TET_FACES = [(0, 1, 2), (1, 0, 3), (0, 2, 3), (2, 1, 3)]
_TRI_FACES = [(0, 1), (1, 2), (2, 0)]


def neumann_rhs(mesh, dofs, face_tables, h_func, boundary_id=3):
    """cell_rhs(i) += scalar_product(h, phi_i) JxW on faces with `boundary_id` (Convergence3D.cpp:309-331, 506-528).

    Returns (dofs, values) with unique dofs, ready for add_rhs().  Host-side: O(boundary faces)."""
    dim = mesh.dim
    faces = _TET_FACES if dim == 3 else _TRI_FACES
    nqf = face_tables.n_qf
    acc = {}
    sel = np.nonzero(mesh.bface_ids == boundary_id)[0]
    for bf in sel:
        cell = mesh.bface_cells[bf]
        cv = mesh.cells[cell]
        fset = set(mesh.bfaces[bf].tolist())
        f = next(k for k, fv in enumerate(faces) if {int(cv[i]) for i in fv} == fset)
        X = mesh.vertices[cv]
        fx = X[list(faces[f])]
        if dim == 3:
            area = 0.5 * np.linalg.norm(np.cross(fx[1] - fx[0], fx[2] - fx[0]))
        else:
            area = np.linalg.norm(fx[1] - fx[0])
        q = slice(f * nqf, (f + 1) * nqf)
        xh = face_tables.points[q]
        xp = X[0] + xh @ (X[1:] - X[0])
        hv = h_func(xp)                      # [nqf, dim]
        N = face_tables.N2[q]                # [nqf, n_p2]
        w = face_tables.weights[q] * area
        loc = np.einsum("q,qa,qc->ac", w, N, hv)   # [n_p2, dim]
        cd = dofs.cell_dofs[cell]
        nv = dim + 1
        for a in range(N.shape[1]):
            base = (dim + 1) * a if a < nv else nv * (dim + 1) + dim * (a - nv)
            for c in range(dim):
                d = int(cd[base + c])
                acc[d] = acc.get(d, 0.0) + loc[a, c]
    keys = np.array(sorted(acc), dtype=np.int32)
    return keys, np.array([acc[k] for k in keys.tolist()])


def velocity_error(mesh, dofs, solution, exact, tables_high, norm="L2"):
    """VectorTools::integrate_difference with the velocity mask + compute_global_error (Convergence3D.cpp:766-794)."""
    dim = mesh.dim
    nv = dim + 1
    X = mesh.vertices[mesh.cells]                        # [nc, nv, dim]
    J = np.transpose(X[:, 1:] - X[:, :1], (0, 2, 1))     # [nc, dim(d), dim(k)] : J[d][k]
    detJ = np.abs(np.linalg.det(J))
    Jinv = np.linalg.inv(J)
    np2 = tables_high.n_p2
    vel_base = [(dim + 1) * a if a < nv else nv * (dim + 1) + dim * (a - nv) for a in range(np2)]
    U = np.stack([solution[dofs.cell_dofs[:, [b + c for b in vel_base]]] for c in range(dim)], axis=2)  # [nc, np2, dim]
    N, dN, w = tables_high.N2, tables_high.dN2, tables_high.weights
    xq = X[:, :1] + np.einsum("qk,ckd->cqd", tables_high.points, X[:, 1:] - X[:, :1])   # [nc, nq, dim]
    uh = np.einsum("qa,cac->cqc", N, U) if False else np.einsum("qa,cad->cqd", N, U)
    nc, nq = xq.shape[0], xq.shape[1]
    ue = exact.velocity(xq.reshape(-1, dim)).reshape(nc, nq, dim)
    err2 = np.einsum("cq,cqd->", detJ[:, None] * w[None, :], (uh - ue) ** 2)
    if norm == "H1":
        gphys = np.einsum("ckd,qak->cqad", Jinv, dN)       # d N_a / d x_d = sum_k Jinv[k][d] dNhat[k]
        guh = np.einsum("cqad,cai->cqid", gphys, U)        # d u_i / d x_d
        ge = exact.gradient(xq.reshape(-1, dim)).reshape(nc, nq, dim, dim)
        err2 += np.einsum("cq,cqid->", detJ[:, None] * w[None, :], (guh - ge) ** 2)
    return float(np.sqrt(err2))


def run_convergence_case(backend_factory, n, prec=0, tol_abs=1e-4, inner_rtol=1e-2, T=3e-4, deltat=4e-4, nu=1e-2):
    """One run of the `convergence` executable on the cube with n cells per side
    (main_convergence3D.cpp:35-52, Convergence3D.cpp:726-764).  `backend_factory(dofs, tables, nu, dt)` returns an
    object with the assemble / add_rhs / apply_boundary_values / solve_time_step / solution interface
    (oracle.Oracle or nsx.Nsx).  Reference quirks kept: Neumann datum evaluated at t_n (Conv:747-750), convection
    assembled twice in the first step (Conv:277,284), error evaluated with the exact solution at T = 3e-4 while the
    state is at t = 4e-4 (Conv:774)."""
    from .frontend import DoFs, Mesh, Tables
    mesh = Mesh.cube(n)
    dofs, tables = DoFs(mesh), Tables(3)
    ftab, htab = Tables(3, Tables.FACE), Tables(3, Tables.HIGH, 5)
    ex = EthierSteinmann(nu)
    be = backend_factory(dofs, tables, nu, deltat)
    X = dofs.support_points
    u0 = np.zeros(dofs.n_dofs)
    ex.set_time(0.0)
    vel = ex.velocity(X[:dofs.n_u])
    u0[:dofs.n_u] = vel[np.arange(dofs.n_u), np.arange(dofs.n_u) % 3]
    u0[dofs.n_u:] = ex.pressure(X[dofs.n_u:])          # VectorTools::interpolate(dof_handler, u_0, ...) (Conv:735)
    if hasattr(be, "set_solution"):
        be.set_solution(u0)
    else:
        be.solution[:] = u0
        be.solution_owned[:] = u0
    time, step = 0.0, 0
    stats = None
    while time < T - 0.5 * deltat:
        ex.set_time(time)                               # function_h.set_time(time) BEFORE the increment (Conv:747-750)
        hd, hv = neumann_rhs(mesh, dofs, ftab, ex.neumann_h, 3)
        time += deltat
        step += 1
        if step == 1:
            be.assemble(1 | 2)                          # TEMAM | DOUBLE_CONVECTION
        else:
            be.assemble_time_step(1)
        be.add_rhs(hd, hv)
        bd, bv = ethier_boundary_values(dofs, ex, time)
        be.apply_boundary_values(bd, bv)
        stats = be.solve_time_step(prec, tol_abs=tol_abs, inner_rtol=inner_rtol)
    sol = np.array(be.solution)
    ex.set_time(T)
    return {"h": 1.0 / (n / 2.0) / 1.0, "L2": velocity_error(mesh, dofs, sol, ex, htab, "L2"),
            "H1": velocity_error(mesh, dofs, sol, ex, htab, "H1"), "stats": stats, "n_dofs": dofs.n_dofs, "solution": sol}


# ------------------------------------------------------------------ forces on the obstacle
def obstacle_faces(mesh, boundary_id=3):
    """(cell, deal.II local face number) of every boundary face with `boundary_id` (the loop NavierStokes3D.cpp:771-786)."""
    faces = _TET_FACES if mesh.dim == 3 else _TRI_FACES
    sel = np.nonzero(mesh.bface_ids == boundary_id)[0]
    cells = mesh.bface_cells[sel].astype(np.int32)
    lf = np.empty(len(sel), dtype=np.int32)
    for k, bf in enumerate(sel):
        cv = mesh.cells[cells[k]]
        fset = set(mesh.bfaces[bf].tolist())
        lf[k] = next(i for i, fv in enumerate(faces) if {int(cv[j]) for j in fv} == fset)
    return cells, lf


def force_coefficients(dim, drag, lift, mean_v, rho=1.0, D=0.1, H=0.41):
    """c_D, c_L: 2F/(rho U^2 D H) in 3D (NavierStokes3D.cpp:835-842), 2F/(U^2 D) in 2D (NavierStokes2D.cpp:849-853)."""
    den = rho * mean_v * mean_v * D * H if dim == 3 else mean_v * mean_v * D
    return 2.0 * drag / den, 2.0 * lift / den


def pressure_difference(mesh, dofs, solution, p_a=None, p_e=None):
    """compute_pressure_difference (NavierStokes3D.cpp:849-923): P1 pressure at two points, p(A) - p(E)."""
    dim = mesh.dim
    p_a = np.array(p_a if p_a is not None else ([0.45, 0.2, 0.205] if dim == 3 else [0.15, 0.2]))
    p_e = np.array(p_e if p_e is not None else ([0.55, 0.2, 0.205] if dim == 3 else [0.25, 0.2]))
    X = mesh.vertices[mesh.cells]
    J = np.transpose(X[:, 1:] - X[:, :1], (0, 2, 1))
    Jinv = np.linalg.inv(J)

    def value(pt):
        xi = np.einsum("ckd,cd->ck", Jinv, pt[None, :] - X[:, 0])
        lam = np.c_[1 - xi.sum(1), xi]
        c = int(np.argmax(lam.min(1)))                   # cell containing the point (largest minimal barycentric)
        if lam[c].min() < -1e-10:
            raise ValueError("point %s is outside the mesh" % pt)
        pd = dofs.cell_dofs[c][[(dim + 1) * v + dim for v in range(dim + 1)]]
        return float(lam[c] @ solution[pd])

    return value(p_a) - value(p_e)
