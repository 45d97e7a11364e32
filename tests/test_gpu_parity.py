"""GPU-vs-oracle parity of the hot path through the C-ABI (tests need a real MI355X)."""
import numpy as np
import pytest

from conftest import Problem, entry_err, record, rel_err

pytestmark = pytest.mark.gpu

# (mesh, dim, level, virtual ranks[, node ordering inside a rank])
# (a 6th entry overrides deltat: BASELINE.json configs[0] runs the 2D cylinder at deltat = 1e-3; the reference's main2D.cpp:21-22 uses 1e-2)
CASES = [("cylinder", 3, 1, 1), ("cylinder", 3, 1, 6), ("cylinder", 2, 2, 1), ("cylinder", 2, 2, 5), ("cube", 3, 3, 1), ("cube", 3, 4, 3),
         ("cylinder", 3, 2, 24, "colour"), ("cylinder", 2, 2, 5, "colour"), ("cylinder", 2, 3, 4, "colour", 1e-3)]


# asserted agreement of (velocity, pressure) after a full step at tol_abs = 1e-11 / inner_rtol = 1e-10, default 1e-10 each; the entries
# below are the cases whose floor is higher, with the reason
TIGHT_BOUND = {}


def _bc(p, time):
    from navierstokes_project_nm4pde_amd.problem import (EthierSteinmann, InletVelocity, cylinder_boundary_values,
                                                         ethier_boundary_values)
    if p.mesh.bface_ids.max() > 3:
        return ethier_boundary_values(p.dofs, EthierSteinmann(p.nu), time)
    return cylinder_boundary_values(p.dofs, InletVelocity(p.dim, 2 if p.dim == 3 else 3), time)


def _case_id(c):
    return "%s%dd-l%d-r%d" % c[:4] + ("-" + c[4] if len(c) > 4 else "") + ("-dt%g" % c[5] if len(c) > 5 else "")


@pytest.fixture(scope="module", params=CASES, ids=_case_id)
def pair(request):
    kind, dim, level, nsub = request.param[:4]
    ordering = request.param[4] if len(request.param) > 4 else "first_touch"
    p = Problem(kind, dim, level, n_sub=nsub, nu=1e-2 if kind == "cube" else 1e-3,
                deltat=request.param[5] if len(request.param) > 5 else (4e-4 if kind == "cube" else None), ordering=ordering)
    p.case_id = _case_id(request.param)
    dev, ora = p.device(), p.oracle()
    u = p.smooth_velocity()
    dev.set_solution(u)
    ora.solution[:] = u
    ora.solution_owned[:] = u
    yield p, dev, ora
    dev.close()


def _matrix_check(p, dev, ora, which, block, name, tol=1e-12):
    """every stored entry against the oracle's: the global maximum norm (1e-12) AND entry by entry (SURVEY 8c pin 4), each entry
    measured against max(|b_ij|, 1e-3 of its row's largest, 1e-4 of the matrix' largest) -- conftest.entry_err"""
    a, b = dev.export_block(which, block), ora.matrix(which, block)
    assert rel_err(a, b) < tol, (name, block)
    e = entry_err(a, b, p.dofs.reference_sparsity(3 if which == 4 else block)[0])
    assert e < 100 * tol, (name, block, e)
    return e


def test_first_assembly_matches_oracle(pair):
    import navierstokes_project_nm4pde_amd.nsx as nsx
    p, dev, ora = pair
    flags = nsx.TEMAM | (nsx.DOUBLE_CONVECTION if p.mesh.bface_ids.max() > 3 else 0)
    dev.assemble(flags)
    ora.assemble(flags)
    worst = 0.0
    for which, name in ((0, "system"), (1, "mass"), (2, "convection"), (3, "stiffness")):
        worst = max(worst, _matrix_check(p, dev, ora, which, 0, name))
    for block in (1, 2):
        worst = max(worst, _matrix_check(p, dev, ora, 0, block, "system"))
    worst = max(worst, _matrix_check(p, dev, ora, 4, 3, "pressure mass"))
    assert rel_err(dev.rhs, ora.rhs) < 1e-12
    record("first_assembly", case=p.case_id, worst_entry_err=worst, rhs=rel_err(dev.rhs, ora.rhs))


def test_dirichlet_matches_oracle(pair):
    p, dev, ora = pair
    bd, bv = _bc(p, p.deltat)
    dev.apply_boundary_values(bd, bv)
    ora.apply_boundary_values(bd, bv)
    for block in (0, 1, 2):
        _matrix_check(p, dev, ora, 0, block, "system after apply_boundary_values")
    assert rel_err(dev.rhs, ora.rhs) < 1e-12
    assert rel_err(dev.solution, ora.solution) < 1e-14


def test_block_vmult_matches_oracle(pair):
    p, dev, ora = pair
    x = np.random.default_rng(7).standard_normal(p.dofs.n_dofs)
    assert rel_err(dev.system_vmult(x), ora.system_vmult(x)) < 1e-13


@pytest.mark.parametrize("prec", [0, 1, 2, 3])
def test_preconditioner_initialize_matches_oracle(pair, prec):
    p, dev, ora = pair
    dev.prec_initialize(prec)
    ora.prec_initialize(prec)
    S_o = ora.schur()
    S_d = dev.schur()
    assert (S_o.indptr == S_d.indptr).all() and (S_o.indices == S_d.indices).all()
    assert rel_err(S_d.data, S_o.data) < 1e-12 and entry_err(S_d.data, S_o.data, S_o.indptr) < 1e-10
    # ILU(0) factors: scalar layout vs the reference's padded layout (same-component entries carry the scalar factor)
    rp, ci, lu = dev.ilu(0)
    g0 = ora.graphs[0]
    luo = ora.ilu_F()
    dim = p.dim
    rows = np.repeat(np.arange(len(g0[0]) - 1), np.diff(g0[0]))
    sel = (rows % dim == 0) & (g0[1] % dim == 0)
    assert sel.sum() == len(lu)
    assert rel_err(lu, luo[sel]) < 1e-11
    assert rel_err(dev.ilu(1)[2], ora.ilu_S(S_o.nnz)) < 1e-10


VMULT_TIGHT_MAX = 8e-14   # measured maximum (round 5, MI355X, 9 cases x 4 preconditioners: 7.7e-14; gpurun_out/parity_maxima.jsonl, "prec_vmult_tight")


@pytest.mark.parametrize("prec", [0, 1, 2, 3])
def test_preconditioner_vmult_tight_matches_oracle(pair, prec):
    p, dev, ora = pair
    dev.prec_initialize(prec)
    ora.prec_initialize(prec)
    src = np.random.default_rng(11).standard_normal(p.dofs.n_dofs)
    yd, sd = dev.prec_vmult(prec, src, inner_rtol=1e-11)
    yo, so = ora.prec_vmult(prec, src, inner_rtol=1e-11)
    assert sd["status"] == 0 and so["status"] == 0
    err = rel_err(yd, yo)
    record("prec_vmult_tight", case=p.case_id, prec=prec, err=err)
    # measured maximum over the 9 cases x 4 preconditioners: see VMULT_TIGHT_MAX below (two inner Krylov solves to 1e-11 in a row, each
    # stopping one iteration earlier or later by rounding); asserted at 100 x that
    assert err < 100 * VMULT_TIGHT_MAX, err


@pytest.mark.parametrize("prec", [0, 3])
def test_time_steps_tight_tolerance(pair, prec):
    """Full steps with tightened tolerances on both sides: the only regime where 1e-10 parity is meaningful."""
    import navierstokes_project_nm4pde_amd.nsx as nsx
    p, dev, ora = pair
    temam_step = nsx.TEMAM if (p.dim == 2 or p.mesh.bface_ids.max() > 3) else 0
    t = p.deltat
    for step in range(2):
        t += p.deltat
        dev.assemble_time_step(temam_step)
        ora.assemble_time_step(temam_step)
        # the two sides assemble from their OWN previous solutions, which agree to the solve tolerance (1e-8 below), not
        # to rounding; assembly from identical input is compared at 1e-12 in test_assemble_time_step_matches_oracle
        assert rel_err(dev.export_block(2, 0), ora.matrix(2, 0)) < 1e-8
        bd, bv = _bc(p, t)
        dev.apply_boundary_values(bd, bv)
        ora.apply_boundary_values(bd, bv)
        assert rel_err(dev.export_block(0, 0), ora.matrix(0, 0)) < 1e-8
        # the absolute tolerance has to stay above the floor the inner solves (1e-10 relative) leave in the preconditioned
        # residual, which scales with the right-hand side ~ 1 / deltat: at deltat = 1e-3 (|rhs| = 7) a run at 1e-11 sits ON that
        # floor and converges at iteration 19 or only after the restart, by rounding (measured: device 19, oracle 32)
        tol = 1e-10 if p.deltat == 1e-3 else 1e-11
        sd = dev.solve_time_step(prec, tol_abs=tol, inner_rtol=1e-10)
        so = ora.solve_time_step(prec, tol_abs=tol, inner_rtol=1e-10)
        assert sd["status"] == 0 and so["status"] == 0
        nu_ = p.dofs.n_u
        xd, xo = dev.solution_owned, np.array(ora.solution_owned)
        err_u = np.abs(xd[:nu_] - xo[:nu_]).max() / np.abs(xo[:nu_]).max()
        err_p = np.abs(xd[nu_:] - xo[nu_:]).max() / np.abs(xo[nu_:]).max()
        record("time_steps_tight", case=p.case_id, prec=prec, step=step, err_u=err_u, err_p=err_p, outer=sd["outer_iterations"])
        # north_star: velocity and pressure within 1e-10 relative with both sides at tightened tolerances (pin 4).  Two
        # solves that stop at a preconditioned residual of `tol` agree to about tol x the conditioning of the preconditioned
        # operator: the bound is asserted per case in TIGHT_BOUND (measured maxima in DESIGN.md section 5)
        bound_u, bound_p = TIGHT_BOUND.get((p.kind, p.dim, prec), (1e-10, 1e-10))
        assert err_u < bound_u and err_p < bound_p, (err_u, err_p)
        # same algorithm, same arithmetic up to rounding: the iteration histories coincide (a restart more or less would
        # show up as tens of iterations)
        for key in ("outer_iterations", "inner_F_iterations", "inner_S_iterations"):
            assert abs(sd[key] - so[key]) <= max(2, 0.05 * so[key]), key


def test_reference_tolerances_iteration_counts(pair):
    """At the reference's own tolerances (1e-4 / 1e-2) the two runs agree to solver tolerance and iterate alike."""
    import navierstokes_project_nm4pde_amd.nsx as nsx
    p, dev, ora = pair
    prec = 0 if p.dim == 3 else 3
    temam_step = nsx.TEMAM if (p.dim == 2 or p.mesh.bface_ids.max() > 3) else 0
    t = 4 * p.deltat
    dev.assemble_time_step(temam_step)
    ora.assemble_time_step(temam_step)
    bd, bv = _bc(p, t)
    dev.apply_boundary_values(bd, bv)
    ora.apply_boundary_values(bd, bv)
    sd = dev.solve_time_step(prec, maxiter=500, check=False)
    so = ora.solve_time_step(prec, maxiter=500)
    assert sd["status"] == 0 and so["status"] == 0
    assert abs(sd["outer_iterations"] - so["outer_iterations"]) <= max(2, 0.2 * so["outer_iterations"])
    scale = np.abs(ora.solution_owned).max()
    assert np.abs(dev.solution_owned - ora.solution_owned).max() / scale < 1e-3


def test_assemble_time_step_matches_oracle(pair):
    """assemble_time_step from IDENTICAL previous solutions: convection matrix, system matrix and right-hand side.
    Runs after the step tests so that it does not change the state they start from."""
    import navierstokes_project_nm4pde_amd.nsx as nsx
    p, dev, ora = pair
    u = p.smooth_velocity(seed=77, amp=0.7)
    dev.set_solution(u)
    ora.solution[:] = u
    ora.solution_owned[:] = u
    flags = nsx.TEMAM if (p.dim == 2 or p.mesh.bface_ids.max() > 3) else 0
    dev.assemble_time_step(flags)
    ora.assemble_time_step(flags)
    assert rel_err(dev.export_block(2, 0), ora.matrix(2, 0)) < 1e-12
    for block in (1, 2):
        assert rel_err(dev.export_block(0, block), ora.matrix(0, block)) < 1e-12
    assert rel_err(dev.rhs, ora.rhs) < 1e-12
    # block (0,0) is compared once the boundary values are applied, as every caller does next (NS3D.cpp:541): until then the
    # reference's constrained rows hold  d - C_old + C_new  from `system -= C_old; system += C_new` (NS3D.cpp:388,512), the
    # library's hold the freshly summed mass + stiffness + convection; apply_boundary_values overwrites both
    bd, bv = _bc(p, 2 * p.deltat)
    dev.apply_boundary_values(bd, bv)
    ora.apply_boundary_values(bd, bv)
    assert rel_err(dev.export_block(0, 0), ora.matrix(0, 0)) < 1e-12
    assert rel_err(dev.rhs, ora.rhs) < 1e-12


@pytest.mark.parametrize("dim", [2, 3])
def test_forces_match_oracle(dim):
    """compute_forces (drag / lift face quadrature on boundary id 3): device kernel vs the oracle's FEFaceValues-style loop."""
    from navierstokes_project_nm4pde_amd.frontend import Tables
    from navierstokes_project_nm4pde_amd.problem import force_coefficients, obstacle_faces, pressure_difference
    p = Problem("cylinder", dim, 2, n_sub=4)
    dev, ora = p.device(), p.oracle()
    u = p.smooth_velocity(seed=99)
    dev.set_solution(u)
    ora.solution[:] = u
    cells, lf = obstacle_faces(p.mesh)
    ftab = Tables(dim, Tables.FACE)
    dev.set_force_faces(cells, lf, ftab)
    fd, fo = dev.compute_forces(), ora.compute_forces(cells, lf, ftab)
    scale = max(abs(fo[0]), abs(fo[1]))
    assert abs(fd[0] - fo[0]) < 1e-12 * scale and abs(fd[1] - fo[1]) < 1e-12 * scale
    cd, cl = force_coefficients(dim, *fd, mean_v=4.0 if dim == 3 else 1.0)
    assert np.isfinite(cd) and np.isfinite(cl)
    assert np.isfinite(pressure_difference(p.mesh, p.dofs, dev.solution))
    dev.close()
