"""Per-rank ILU(0) with FEW, LARGE ranks — R = 1 is the serial reference (Preconditioners.hpp:215-216 with one MPI rank),
R = 8 one rank per GPU — goes through the level-per-launch kernels (k_ilu_factor_level / k_ilu_solve_level) instead of
the wave-per-block stream.  Parity of that path against the oracle (tests need a real MI355X)."""
import os

import numpy as np
import pytest

from conftest import Problem, rel_err

pytestmark = pytest.mark.gpu


def _bc(p, time):
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    return cylinder_boundary_values(p.dofs, InletVelocity(p.dim, 2 if p.dim == 3 else 3), time)


@pytest.fixture()
def forced_levelled():
    """Force every block above 64 rows onto the levelled path (the threshold is read when the schedules are built)."""
    old = os.environ.get("NSX_LEVELLED_MIN")
    os.environ["NSX_LEVELLED_MIN"] = "64"
    yield
    if old is None:
        os.environ.pop("NSX_LEVELLED_MIN", None)
    else:
        os.environ["NSX_LEVELLED_MIN"] = old


def _scalar_factor_of_oracle(p, ora, n_lu):
    g0 = ora.graphs[0]
    rows = np.repeat(np.arange(len(g0[0]) - 1), np.diff(g0[0]))
    sel = (rows % p.dim == 0) & (g0[1] % p.dim == 0)
    assert sel.sum() == n_lu
    return ora.ilu_F()[sel]


@pytest.mark.parametrize("case", [("cylinder", 3, 1, 1, "first_touch"), ("cylinder", 3, 1, 3, "colour"), ("cylinder", 2, 2, 2, "first_touch"),
                                  ("cylinder", 3, 1, 2, "colour_all")],   # pressure nodes by colour of the Schur graph as well
                         ids=lambda c: "%s%dd-l%d-r%d-%s" % c)
def test_levelled_factor_solve_and_step_match_oracle(forced_levelled, case):
    import oracle
    import navierstokes_project_nm4pde_amd.nsx as nsx
    kind, dim, level, nsub, ordering = case
    p = Problem(kind, dim, level, n_sub=nsub, ordering=ordering)
    dev, ora = p.device(), p.oracle()
    u = p.smooth_velocity()
    dev.set_solution(u)
    ora.solution[:] = u
    ora.solution_owned[:] = u
    for o in (dev, ora):
        o.assemble(nsx.TEMAM)
        o.apply_boundary_values(*_bc(p, p.deltat))
    prec = 0 if dim == 3 else 3
    dev.prec_initialize(prec)
    ora.prec_initialize(prec)
    rp, ci, lu = dev.ilu(0)
    assert rel_err(lu, _scalar_factor_of_oracle(p, ora, len(lu))) < 1e-11
    S_o = ora.schur()
    rps, cis, lus = dev.ilu(1)
    assert rel_err(lus, ora.ilu_S(S_o.nnz)) < 1e-10
    # the triangular solves alone: z = (LDU)^-1 b per rank block, all components
    rng = np.random.default_rng(5)
    b = rng.standard_normal(p.dofs.n_u)
    bptr = np.asarray(p.dofs.owned_u_ptr) if nsub > 1 else np.array([0, p.dofs.n_u // dim])
    z = dev.ilu_apply(0, b)
    for c in range(dim):
        zo = oracle.ilu0_solve(rp, ci, lu, bptr, b[c::dim])
        assert rel_err(z[c::dim], zo) < 1e-11
    bp = rng.standard_normal(p.dofs.n_p)
    pptr = np.asarray(p.dofs.owned_p_ptr) if nsub > 1 else np.array([0, p.dofs.n_p])
    assert rel_err(dev.ilu_apply(1, bp), oracle.ilu0_solve(rps, cis, lus, pptr, bp)) < 1e-10
    # a full step with tightened tolerances
    sd = dev.solve_time_step(prec, tol_abs=1e-11, inner_rtol=1e-10)
    so = ora.solve_time_step(prec, tol_abs=1e-11, inner_rtol=1e-10)
    assert sd["status"] == 0 and so["status"] == 0
    assert rel_err(dev.solution_owned, ora.solution_owned) < 1e-8
    for key in ("outer_iterations", "inner_F_iterations", "inner_S_iterations"):
        assert abs(sd[key] - so[key]) <= max(2, 0.05 * so[key]), key
    dev.close()


def test_serial_reference_layout_level3():
    """R = 1, first-touch numbering, 131 403 DoF: the layout of the reference run without mpirun.  Factors and one time
    step at the reference's own tolerances (iteration counts and solution to solver tolerance)."""
    import navierstokes_project_nm4pde_amd.nsx as nsx
    p = Problem("cylinder", 3, 3, n_sub=1, ordering="first_touch")
    assert p.dofs.n_dofs == 131403
    dev, ora = p.device(), p.oracle()
    u = p.smooth_velocity()
    dev.set_solution(u)
    ora.solution[:] = u
    ora.solution_owned[:] = u
    for o in (dev, ora):
        o.assemble(nsx.TEMAM)
        o.apply_boundary_values(*_bc(p, p.deltat))
    dev.prec_initialize(0)
    ora.prec_initialize(0)
    rp, ci, lu = dev.ilu(0)
    assert rel_err(lu, _scalar_factor_of_oracle(p, ora, len(lu))) < 1e-10
    assert rel_err(dev.ilu(1)[2], ora.ilu_S(ora.schur().nnz)) < 1e-9
    sd = dev.solve_time_step(0)
    so = ora.solve_time_step(0)
    assert sd["status"] == 0 and so["status"] == 0
    assert abs(sd["outer_iterations"] - so["outer_iterations"]) <= max(2, 0.2 * so["outer_iterations"])
    assert abs(sd["inner_F_iterations"] - so["inner_F_iterations"]) <= max(4, 0.2 * so["inner_F_iterations"])
    assert rel_err(dev.solution_owned, ora.solution_owned) < 1e-3
    dev.close()
