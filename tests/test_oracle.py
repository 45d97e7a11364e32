"""CPU suite: the oracle (CPU restatement of the reference algorithm) against its pins.

The reference ships no golden vectors (parity unpinned), so the pins are
  * exact element integrals (tests/golden/p2p1_element_dim{2,3}.json, SymPy rationals, tools/gen_golden.py),
  * polynomial exactness / structure identities on multi-cell meshes,
  * the Ethier-Steinmann manufactured solution of the reference's `convergence` executable,
  * independent SciPy re-computations of ILU(0), the Schur product and the block mat-vec.
"""
import json
import os
from fractions import Fraction

import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from conftest import Problem, rel_err
from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _frac(x):
    return float(Fraction(x))


def _single_cell(tmp_path, verts):
    dim = len(verts[0])
    p = tmp_path / "cell.msh"
    with open(p, "w") as f:
        f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % len(verts))
        for i, v in enumerate(verts):
            xyz = list(v) + [0.0] * (3 - dim)
            f.write("%d %.17g %.17g %.17g\n" % (i + 1, *xyz))
        f.write("$EndNodes\n$Elements\n1\n")
        f.write("1 %d 2 10 1 %s\n" % (4 if dim == 3 else 2, " ".join(str(i + 1) for i in range(dim + 1))))
        f.write("$EndElements\n")
    return Mesh.read_msh(p)


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("case", ["reference", "affine"])
def test_element_matrices_match_closed_forms(tmp_path, dim, case):
    g = json.load(open(os.path.join(GOLDEN, "p2p1_element_dim%d.json" % dim)))[case]
    verts = [[_frac(c) for c in v] for v in g["vertices"]]
    mesh = _single_cell(tmp_path, verts)
    assert (mesh.cells[0] == np.arange(dim + 1)).all()          # positively oriented as written
    dofs, tables = DoFs(mesh), Tables(dim)
    n2, n1 = tables.n_p2, tables.n_p1
    o = oracle.Oracle(dofs, tables, 1.0, 1.0)                    # nu = dt = 1: mass/dt = mass, nu*K = K
    w = np.array([[_frac(c) for c in r] for r in g["w"]])
    o.solution[:dofs.n_u] = w.ravel()                            # single cell: node a = local index a, dof = dim*a + c
    o.assemble(oracle.TEMAM)

    def scalar(which):
        M = o.scipy(which, 0).toarray()
        S = M[0::dim, 0::dim]
        for c in range(1, dim):                                  # delta_cd (x) scalar structure (SURVEY A-struct iii)
            assert np.allclose(M[c::dim, c::dim], S, rtol=0, atol=1e-15 * abs(S).max())
            assert abs(M[c::dim, 0::dim]).max() == 0.0
        return S

    tol = 2e-14
    exact = lambda key: np.array([[_frac(x) for x in r] for r in g[key]])
    assert rel_err(scalar(1), exact("mass")) < tol
    assert rel_err(scalar(3), exact("stiffness")) < tol
    assert rel_err(scalar(2), exact("conv") + 0.5 * exact("temam")) < tol
    D = np.array([[[_frac(x) for x in v] for v in a] for a in g["div"]])        # [a][v][c]
    G = o.scipy(0, 1).toarray().reshape(n2, dim, n1)                             # block(0,1)[(a,c), v] = -D
    B = o.scipy(0, 2).toarray().reshape(n1, n2, dim)                             # block(1,0)[v, (a,c)] = +D
    assert rel_err(G, -np.transpose(D, (0, 2, 1))) < tol
    assert rel_err(B, np.transpose(D, (1, 0, 2))) < tol
    assert rel_err(o.scipy(4, 3).toarray(), exact("pmass")) < tol


@pytest.mark.parametrize("dim", [2, 3])
def test_quadrature_exact_to_degree_5(dim):
    from math import factorial
    t = Tables(dim)
    for e in np.ndindex(*([6] * dim)):
        if sum(e) > 5:
            continue
        num = np.sum(t.weights * np.prod(t.points ** np.array(e), axis=1))
        ex = np.prod([factorial(k) for k in e]) / factorial(sum(e) + dim)
        assert abs(num - ex) < 1e-16 + 1e-14 * ex
    assert np.allclose(t.N2.sum(1), 1) and np.allclose(t.N1.sum(1), 1) and np.allclose(t.dN2.sum(1), 0)


@pytest.mark.parametrize("kind,dim", [("box", 2), ("box", 3), ("cylinder", 2), ("cylinder", 3)])
def test_assembled_operators_polynomial_identities(kind, dim):
    p = Problem(kind, dim, 1, nu=0.1, deltat=0.01)
    d, o = p.dofs, p.oracle()
    X = d.support_points
    u = np.zeros(d.n_dofs)
    for c in range(dim):
        Xc = X[c:d.n_u:dim]
        u[c:d.n_u:dim] = 1.0 + 0.5 * Xc[:, 0] ** 2 - (c + 1) * Xc[:, 1] * Xc[:, dim - 1]
    o.solution[:] = u
    o.assemble(0)
    M, K, C = o.scipy(1, 0), o.scipy(3, 0), o.scipy(2, 0)
    G, B, Mp = o.scipy(0, 1), o.scipy(0, 2), o.scipy(4, 3)
    Xm = p.mesh.vertices[p.mesh.cells]
    vol = abs(np.linalg.det(Xm[:, 1:] - Xm[:, :1])).sum() / (2 if dim == 2 else 6)
    assert abs(M.sum() * p.deltat - dim * vol) < 1e-12 * vol            # sum of mass = dim * |Omega|
    assert abs(Mp.sum() * p.nu - vol) < 1e-12 * vol
    assert abs(K.sum(1)).max() < 1e-12 * abs(K).max()                   # constants are in the kernel of the stiffness
    assert abs(C @ np.ones(d.n_u)).max() < 1e-12 * abs(C).max()         # (w . grad) 1 = 0
    assert abs(G + B.T).max() == 0.0                                    # block(0,1) = -block(1,0)^T exactly
    assert rel_err(o.rhs[:d.n_u], M @ u[:d.n_u]) < 1e-13                # rhs = (M/dt) u_n
    assert abs(o.rhs[d.n_u:]).max() == 0.0
    ul = np.zeros(d.n_u)                                                # B u for a linear field = int psi div u
    for c in range(dim):
        ul[c::dim] = (c + 1) * X[c:d.n_u:dim, c]
    assert abs((B @ ul).sum() - vol * sum(range(1, dim + 1))) < 1e-12 * vol
    # energy identity of the convective form with Temam: x^T (C + 1/2 div-term) x = 1/2 int_boundary (w.n) |x|^2 ; for a
    # field x vanishing on the boundary the skew part is all that is left
    o2 = p.oracle()
    o2.solution[:] = u
    o2.assemble(oracle.TEMAM)
    Ct = o2.scipy(2, 0)
    x = np.random.default_rng(3).standard_normal(d.n_u)
    bd = np.unique(np.concatenate([d.boundary_dofs(b) for b in np.unique(p.mesh.bface_ids)]))
    x[bd] = 0.0
    assert abs(x @ (Ct @ x)) < 1e-10 * (abs(Ct) @ abs(x)) @ abs(x)
    # system(0,0) = M + K + C and is the same scalar operator on every component
    assert abs((M + K + C) - o.scipy(0, 0)).max() < 1e-14 * abs(M).max()


def test_time_step_assembly_equals_fresh_assembly():
    """F - C_old + C_new of assemble_time_step (NS3D.cpp:388,512) equals a first assembly at the new state."""
    p = Problem("cylinder", 3, 1)
    o1, o2 = p.oracle(), p.oracle()
    u1, u2 = p.smooth_velocity(1), p.smooth_velocity(2)
    o1.solution[:] = u1
    o1.assemble(0)
    o1.solution[:] = u2
    o1.assemble_time_step(0)
    o2.solution[:] = u2
    o2.assemble(0)
    assert rel_err(o1.matrix(0, 0), o2.matrix(0, 0)) < 1e-13
    assert rel_err(o1.matrix(2, 0), o2.matrix(2, 0)) < 1e-15
    assert rel_err(o1.rhs, o2.rhs) < 1e-15


def test_dirichlet_rows():
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    p = Problem("cylinder", 3, 1, n_sub=3)
    d, o = p.dofs, p.oracle()
    o.solution[:] = p.smooth_velocity()
    o.assemble(oracle.TEMAM)
    F0 = o.scipy(0, 0)
    bd, bv = cylinder_boundary_values(d, InletVelocity(3), 2e-4)
    o.apply_boundary_values(bd, bv)
    F, G, B = o.scipy(0, 0), o.scipy(0, 1), o.scipy(0, 2)
    ranks = np.searchsorted(3 * d.owned_u_ptr, bd, side="right") - 1
    dbar = np.array([abs(F0[3 * d.owned_u_ptr[r], 3 * d.owned_u_ptr[r]]) for r in range(d.n_subdomains)])
    rows = F[bd]
    assert np.allclose(rows.diagonal(k=0) if False else np.array([F[i, i] for i in bd]), dbar[ranks], rtol=1e-15)
    assert abs(rows.sum(1).A1 - dbar[ranks]).max() == 0.0               # nothing but the diagonal left
    assert abs(G[bd]).max() == 0.0                                      # off-diagonal block row cleared
    assert rel_err(o.rhs[bd], dbar[ranks] * bv) < 1e-15
    assert (o.solution[bd] == bv).all()
    free = np.setdiff1d(np.arange(d.n_u), bd)
    assert abs(F[free] - F0[free]).max() == 0.0                         # columns are NOT eliminated
    assert abs(B - o.scipy(0, 2)).max() == 0.0
    inlet_x = bd[(np.isin(bd, d.boundary_dofs(0))) & (bd % 3 == 0)]
    assert bv[np.isin(bd, inlet_x)].max() > 1.0                         # the parabolic profile is really applied


def _ilu0_reference(A, blocks):
    """Textbook IKJ ILU(0) per diagonal block (dense work row), Ifpack storage convention."""
    A = A.tocsr()
    out = A.copy().astype(float)
    out.data[:] = 0
    for b in range(len(blocks) - 1):
        r0, r1 = blocks[b], blocks[b + 1]
        sub = A[r0:r1, r0:r1].toarray()
        pat = sub != 0
        pat |= np.eye(r1 - r0, dtype=bool)
        patA = (A[r0:r1, r0:r1] != 0).toarray() | (abs(A[r0:r1, r0:r1]).toarray() >= 0) & (A[r0:r1, r0:r1].toarray() != 0)
        struct = np.zeros_like(pat)
        Ab = A[r0:r1, r0:r1].tocsr()
        for i in range(r1 - r0):
            struct[i, Ab.indices[Ab.indptr[i]:Ab.indptr[i + 1]]] = True
        LU = sub.copy()
        n = r1 - r0
        for i in range(1, n):
            for k in range(i):
                if not struct[i, k]:
                    continue
                LU[i, k] /= LU[k, k]
                js = np.nonzero(struct[i, k + 1:])[0] + k + 1
                LU[i, js] -= LU[i, k] * LU[k, js] * struct[k, js]
        d = np.diag(LU).copy()
        U = np.triu(LU, 1) / d[:, None]
        L = np.tril(LU, -1)
        full = L + U + np.diag(1.0 / d)
        for i in range(n):
            cols = Ab.indices[Ab.indptr[i]:Ab.indptr[i + 1]]
            gpos = np.arange(A.indptr[r0 + i], A.indptr[r0 + i + 1])
            gcols = A.indices[gpos]
            inb = (gcols >= r0) & (gcols < r1)
            out.data[gpos[inb]] = full[i, gcols[inb] - r0]
    return out


def test_ilu0_factor_and_solve_against_dense_recomputation():
    p = Problem("cylinder", 2, 1)
    o = p.oracle()
    o.solution[:] = p.smooth_velocity()
    o.assemble(oracle.TEMAM)
    A = o.scipy(0, 0)
    n = A.shape[0]
    blocks = np.array([0, n // 3 - (n // 3) % 2, 2 * (n // 3) - (2 * (n // 3)) % 2, n], dtype=np.int32)
    lu = oracle.ilu0_factor(A.indptr, A.indices, A.data, blocks)
    ref = _ilu0_reference(A, blocks)
    assert rel_err(lu, ref.data) < 1e-12
    b = np.random.default_rng(5).standard_normal(n)
    x = oracle.ilu0_solve(A.indptr, A.indices, lu, blocks, b)
    # dense check of U^{-1} D^{-1} L^{-1} b per block
    for k in range(3):
        r0, r1 = blocks[k], blocks[k + 1]
        M = sp.csr_matrix((lu, A.indices, A.indptr), shape=A.shape)[r0:r1, r0:r1].toarray()
        L = np.tril(M, -1) + np.eye(r1 - r0)
        U = np.triu(M, 1) + np.eye(r1 - r0)
        y = np.linalg.solve(U, np.diag(M) * np.linalg.solve(L, b[r0:r1]))
        assert rel_err(x[r0:r1], y) < 1e-11
    # ILU(0) of the padded reference matrix == scalar ILU(0) replicated on the components
    S = A[0::2, 0::2].tocsr()
    S.sort_indices()
    lus = oracle.ilu0_factor(S.indptr, S.indices, S.data, blocks // 2)
    Lp = sp.csr_matrix((lu, A.indices, A.indptr), shape=A.shape)
    assert rel_err(Lp[0::2, 0::2].tocsr().data[Lp[0::2, 0::2].tocsr().data != 0], lus[lus != 0]) < 1e-13
    assert abs(Lp[0::2, 1::2]).max() == 0.0


@pytest.mark.parametrize("prec", [0, 1, 2, 3])
def test_schur_complement_is_B_D_Bt(prec):
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    p = Problem("cylinder", 2, 1)
    d, o = p.dofs, p.oracle()
    o.solution[:] = p.smooth_velocity()
    o.assemble(oracle.TEMAM)
    bd, bv = cylinder_boundary_values(d, InletVelocity(2, 3), 1e-2)
    o.apply_boundary_values(bd, bv)
    o.prec_initialize(prec)
    F, G, B, M = o.scipy(0, 0), o.scipy(0, 1), o.scipy(0, 2), o.scipy(1, 0)
    if prec == 0:
        V = -1.0 / M.diagonal()
    elif prec in (1, 3):
        V = -1.0 / F.diagonal()
    else:
        V = -1.0 / np.asarray(abs(M).sum(1)).ravel()
    S = (B @ sp.diags(V) @ G).toarray()
    assert rel_err(o.schur().toarray(), S) < 1e-13
    ev = np.linalg.eigvalsh(0.5 * (S + S.T))
    assert ev.min() > -1e-10 * ev.max()                                # negative_S = +B D^-1 B^T is symmetric positive semi-definite


@pytest.mark.parametrize("prec", [0, 1, 2, 3])
def test_preconditioned_gmres_solves_the_saddle_point_system(prec):
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    p = Problem("cylinder", 2, 1, n_sub=2)
    d, o = p.dofs, p.oracle()
    o.assemble(oracle.TEMAM)
    bd, bv = cylinder_boundary_values(d, InletVelocity(2, 3), 1e-2)
    o.apply_boundary_values(bd, bv)
    A = sp.bmat([[o.scipy(0, 0), o.scipy(0, 1)], [o.scipy(0, 2), None]]).tocsr()
    b = o.rhs.copy()
    st = o.solve_time_step(prec, tol_abs=1e-11, inner_rtol=1e-10)
    assert st["status"] == 0
    x = np.array(o.solution_owned)
    assert np.linalg.norm(A @ x - b) < 1e-8 * np.linalg.norm(b)
    assert rel_err(o.system_vmult(x), A @ x) < 1e-13
    assert rel_err(x[bd], bv) < 1e-9                                    # Dirichlet values reproduced by the solve
    assert (np.array(o.solution) == x).all()                           # solution = solution_owned (NS3D.cpp:638)


def test_ethier_steinmann_convergence_rates():
    """The reference's only known-answer run (main_convergence3D.cpp:14-73): one implicit Euler step of the
    Ethier-Steinmann flow; velocity L2 error O(h^3), H1 error O(h^2)."""
    from navierstokes_project_nm4pde_amd.problem import run_convergence_case
    res = [run_convergence_case(lambda d, t, nu, dt: oracle.Oracle(d, t, nu, dt), n, tol_abs=1e-9, inner_rtol=1e-6) for n in (2, 4, 8)]
    l2 = [r["L2"] for r in res]
    h1 = [r["H1"] for r in res]
    rate_l2 = [np.log2(l2[i] / l2[i + 1]) for i in range(2)]
    rate_h1 = [np.log2(h1[i] / h1[i + 1]) for i in range(2)]
    # the doubled convective term of the first assembly (Convergence3D.cpp:277,284) is an O(dt) consistency error
    # (~4e-4 * |u.grad u|), a floor the L2 error approaches on the finest mesh: the rate bends from 2.9 to 2.6
    assert rate_l2[0] > 2.8 and rate_l2[1] > 2.5 and min(rate_h1) > 1.85, (l2, h1)
    assert l2[-1] < 6e-3 and h1[-1] < 0.15


@pytest.mark.parametrize("dim", [2, 3])
def test_forces_hydrostatic_known_answer(dim):
    """compute_forces on u = 0, p = x: drag = -int p n_x dS over the obstacle with n pointing into the fluid = -|obstacle|
    (divergence theorem on the polygonal cylinder), lift = 0."""
    from navierstokes_project_nm4pde_amd.problem import obstacle_faces
    p = Problem("cylinder", dim, 2)
    d, o = p.dofs, p.oracle()
    o.solution[:] = 0.0
    o.solution[d.n_u:] = d.support_points[d.n_u:, 0]
    cells, lf = obstacle_faces(p.mesh)
    ftab = Tables(dim, Tables.FACE)
    drag, lift = o.compute_forces(cells, lf, ftab)
    Xm = p.mesh.vertices[p.mesh.cells]
    fluid = abs(np.linalg.det(Xm[:, 1:] - Xm[:, :1])).sum() / (2 if dim == 2 else 6)
    box = 2.2 * 0.41 if dim == 2 else 2.5 * 0.41 * 0.41
    assert abs(drag + (box - fluid)) < 1e-12 and abs(lift) < 1e-12
    # a rigid shear u = (y, 0, 0): grad u = e_x (x) e_y, so the 2D stress force is nu * int n_y dS e_x = 0 on a closed curve
    o.solution[:] = 0.0
    o.solution[0:d.n_u:dim] = d.support_points[0:d.n_u:dim, 1]
    drag, lift = o.compute_forces(cells, lf, ftab)
    assert abs(drag) < 1e-12 and abs(lift) < 1e-12


def test_threaded_oracle_build_equals_the_serial_restatement():
    """oracle/liboracle_mt.so (bench.py's all-cores CPU baseline: the same source with -fopenmp) against liboracle.so:
    same iteration history, solution equal to summation-order rounding."""
    import oracle
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    p = Problem("cylinder", 3, 1, n_sub=4)
    out = []
    for threads in (1, 4):
        o = oracle.Oracle(p.dofs, p.tables, p.nu, p.deltat, threads=threads)
        assert o.threads == threads
        inlet = InletVelocity(3)
        o.assemble(oracle.TEMAM)
        o.apply_boundary_values(*cylinder_boundary_values(p.dofs, inlet, p.deltat))
        st1 = o.solve_time_step(oracle.YOSIDA, tol_abs=1e-10, inner_rtol=1e-8)
        o.assemble_time_step(0)
        o.apply_boundary_values(*cylinder_boundary_values(p.dofs, inlet, 2 * p.deltat))
        st2 = o.solve_time_step(oracle.YOSIDA, tol_abs=1e-10, inner_rtol=1e-8)
        out.append((st1, st2, o.solution_owned.copy(), o.matrix(0, 0).copy()))
    (a1, a2, xa, Fa), (b1, b2, xb, Fb) = out
    assert a1["status"] == 0 and b1["status"] == 0 and a2["status"] == 0 and b2["status"] == 0
    assert rel_err(Fb, Fa) < 1e-13
    assert abs(a2["outer_iterations"] - b2["outer_iterations"]) <= 1
    assert rel_err(xb, xa) < 1e-7


def test_bench_state_transfer_between_numberings():
    """bench.py hands the GPU run's state to the CPU baseline, which numbers the same mesh with other ranks / node order."""
    import bench
    _, da, _ = bench.build_problem(1, 8, 1, "colour")
    _, db, _ = bench.build_problem(1, 3, 1, "first_touch")
    X = da.support_points
    x = np.sin(3 * X[:, 0]) + 2 * X[:, 1] ** 2 - X[:, 2] + 0.25 * (np.arange(da.n_dofs) % 3) * (np.arange(da.n_dofs) < da.n_u)
    y = bench.transfer_state(da, x, db)
    Y = db.support_points
    expect = np.sin(3 * Y[:, 0]) + 2 * Y[:, 1] ** 2 - Y[:, 2] + 0.25 * (np.arange(db.n_dofs) % 3) * (np.arange(db.n_dofs) < db.n_u)
    assert np.array_equal(y, expect)


@pytest.mark.parametrize("dim,prec", [(3, 0), (2, 3), (3, 2)])
def test_compact_storage_mode_equals_the_reference_shaped_layout(dim, prec):
    """orc_set_compact (bench.py's "best CPU" baseline): products with system(0,0) and its per-rank ILU(0) on the scalar P2
    operator instead of the reference's padded dim x dim couplings -- the same algorithm, so the same iteration history and
    the same solution up to rounding (the cross-component fill of the padded ILU(0) is exactly zero)."""
    import oracle
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    p = Problem("cylinder", dim, 1 if dim == 3 else 2, n_sub=3)
    out = []
    for compact in (False, True):
        o = oracle.Oracle(p.dofs, p.tables, p.nu, p.deltat, compact=compact)
        u = p.smooth_velocity()
        o.solution[:] = u
        o.solution_owned[:] = u
        o.assemble(oracle.TEMAM)
        o.apply_boundary_values(*cylinder_boundary_values(p.dofs, InletVelocity(dim, 2 if dim == 3 else 3), p.deltat))
        x = np.random.default_rng(3).standard_normal(p.dofs.n_dofs)
        y = o.system_vmult(x) if not compact else None
        st = o.solve_time_step(prec, tol_abs=1e-10, inner_rtol=1e-9)
        yc = o.system_vmult(x)                                  # (compact: the scalar operator exists after the first initialize)
        out.append((st, np.array(o.solution_owned), y if y is not None else yc))
    (s0, x0, y0), (s1, x1, y1) = out
    assert s0["status"] == 0 and s1["status"] == 0
    for key in ("outer_iterations", "inner_F_iterations", "inner_S_iterations"):
        assert abs(s0[key] - s1[key]) <= max(1, 0.02 * s0[key]), key
    assert np.abs(x0 - x1).max() < 1e-8 * np.abs(x0).max()
    assert np.abs(y0 - y1).max() < 1e-13 * np.abs(y0).max()
