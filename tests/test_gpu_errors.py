"""GPU suite: error behaviour of the C-ABI (what the reference throws) and the edge cases of its inputs."""
import os

import numpy as np
import pytest

from conftest import Problem, rel_err

pytestmark = pytest.mark.gpu


def _bc(p, time):
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    return cylinder_boundary_values(p.dofs, InletVelocity(p.dim, 2), time)


@pytest.fixture(scope="module")
def prob():
    return Problem("cylinder", 3, 1, n_sub=4, ordering="colour")


def _assembled(p):
    import navierstokes_project_nm4pde_amd.nsx as nsx
    dev, ora = p.device(), p.oracle()
    u = p.smooth_velocity()
    dev.set_solution(u)
    ora.solution[:] = u
    ora.solution_owned[:] = u
    dev.assemble(nsx.TEMAM)
    ora.assemble(nsx.TEMAM)
    bd, bv = _bc(p, p.deltat)
    dev.apply_boundary_values(bd, bv)
    ora.apply_boundary_values(bd, bv)
    return dev, ora


def test_call_order_and_argument_errors(prob):
    import navierstokes_project_nm4pde_amd.nsx as nsx
    dev = prob.device()
    with pytest.raises(nsx.NsxError) as e:          # assemble_time_step needs the first-step operators
        dev.assemble_time_step(0)
    assert e.value.code == -1
    with pytest.raises(nsx.NsxError) as e:          # no system to constrain yet
        dev.apply_boundary_values(*_bc(prob, prob.deltat))
    assert e.value.code == -1
    dev.set_solution(prob.smooth_velocity())
    dev.assemble(nsx.TEMAM)
    bd, bv = _bc(prob, prob.deltat)
    with pytest.raises(nsx.NsxError) as e:          # std::map order is part of the contract
        dev.apply_boundary_values(bd[::-1].copy(), bv[::-1].copy())
    assert e.value.code == -1
    with pytest.raises(nsx.NsxError) as e:          # ComponentMask of the reference: velocity only (NS3D.cpp:337-338)
        dev.apply_boundary_values(np.array([prob.dofs.n_u + 1], dtype=np.int32), np.array([1.0]))
    assert e.value.code == -3
    dev.apply_boundary_values(bd, bv)
    with pytest.raises(nsx.NsxError) as e:          # reference: std::runtime_error("Invalid preconditioner type"), NS3D.cpp:633
        dev.solve_time_step(7)
    assert e.value.code == -1 and "Invalid preconditioner type" in str(e.value)
    dev.close()


def test_empty_boundary_map_changes_nothing(prob):
    import navierstokes_project_nm4pde_amd.nsx as nsx
    dev, ora = prob.device(), prob.oracle()
    u = prob.smooth_velocity()
    dev.set_solution(u)
    ora.solution[:] = u
    dev.assemble(nsx.TEMAM)
    ora.assemble(nsx.TEMAM)
    before = dev.export_block(0, 0).copy(), dev.rhs.copy()
    dev.apply_boundary_values(np.zeros(0, dtype=np.int32), np.zeros(0))
    ora.apply_boundary_values(np.zeros(0, dtype=np.int32), np.zeros(0))
    assert (dev.export_block(0, 0) == before[0]).all() and (dev.rhs == before[1]).all()
    assert rel_err(dev.export_block(0, 0), ora.matrix(0, 0)) < 1e-12 and rel_err(dev.export_block(0, 1), ora.matrix(0, 1)) < 1e-12
    dev.close()


def test_outer_no_convergence_is_reported_like_solver_control(prob):
    """SolverControl::NoConvergence after maxiter steps (NS3D.cpp:553): same step count and residual as the oracle."""
    import navierstokes_project_nm4pde_amd.nsx as nsx
    dev, ora = _assembled(prob)
    so = ora.solve_time_step(0, maxiter=5)
    with pytest.raises(nsx.NsxError) as e:
        dev.solve_time_step(0, maxiter=5)
    assert e.value.code == -4 and "outer GMRES did not converge" in str(e.value)
    dev2, _ = _assembled(prob)
    sd = dev2.solve_time_step(0, maxiter=5, check=False)
    assert sd["status"] == so["status"] == 1 and sd["outer_iterations"] == so["outer_iterations"] == 5
    assert abs(sd["final_residual"] - so["final_residual"]) < 1e-6 * so["final_residual"]
    dev.close()
    dev2.close()


def test_inner_no_convergence_is_reported(prob):
    dev, ora = _assembled(prob)
    sd = dev.solve_time_step(0, inner_maxiter=2, maxiter=3, check=False)
    so = ora.solve_time_step(0, inner_maxiter=2, maxiter=3)
    assert sd["status"] != 0 and so["status"] != 0 and sd["status"] == so["status"]
    dev.close()


def test_unsupported_quadrature_size_fails_loudly(prob):
    import navierstokes_project_nm4pde_amd.nsx as nsx
    from navierstokes_project_nm4pde_amd.frontend import Tables
    t = Tables(3, rule=2, order=5)                   # a conical rule the cell kernels are not instantiated for
    assert t.n_q not in (4, 10, 11, 14, 15, 24)
    with pytest.raises(nsx.NsxError) as e:           # at the latest when the first cell kernel is asked for
        dev = nsx.Nsx(prob.dofs, t, 1e-3, 2e-4)
        dev.set_solution(prob.smooth_velocity())
        dev.assemble(nsx.TEMAM)
    assert e.value.code == -3


def test_gram_schmidt_as_one_launch_equals_the_launch_per_link_chain(prob):
    """NSX_MGS=0 runs SolverGMRES' add_and_dot chain as separate launches; NSX_MGS_LINKS=1 the same chain link by link in one
    persistent launch (same arithmetic per entry); NSX_MGS_LINKS=m (2..5) evaluates m links per grid-wide exchange by linearity
    of the dot product: h_j = v_j.w_j0 - sum_i (v_i.v_j) h_i, identical in exact arithmetic whatever the basis, so the
    coefficients differ by the rounding of the sums only; NSX_MGS_LINKS=0 (the default) ALL links in one exchange with the basis'
    Gram matrix kept on the device, and |w|^2 after the sweep from the same numbers.  All must walk through the same history and
    end at the same vector."""
    res = []
    # NSX_MGS_MAXWG=1: a resident grid of one workgroup cannot hold the vector (> 20 entries per thread), which is what a mesh of
    # several million DoF does to the real grid: the sweep then runs as two passes (all dot products + the Gram row, then the updates)
    cases = [{"NSX_MGS": "0"}, {"NSX_MGS_LINKS": "0"}, {"NSX_MGS_LINKS": "1"}, {"NSX_MGS_LINKS": "2"}, {"NSX_MGS_LINKS": "3"}, {"NSX_MGS_LINKS": "4"},
             {"NSX_MGS_LINKS": "5"}, {"NSX_MGS_MAXWG": "1"}]
    for env in cases:
        os.environ.update(env)
        try:
            dev, _ = _assembled(prob)
            st = dev.solve_time_step(3, tol_abs=1e-10, inner_rtol=1e-8)
            res.append((st, dev.solution_owned.copy()))
            dev.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    s0, x0 = res[0]
    assert s0["status"] == 0 and s0["outer_iterations"] > 5
    for s1, x1 in res[1:]:
        for key in ("outer_iterations", "inner_F_iterations", "inner_S_iterations"):
            assert abs(s0[key] - s1[key]) <= max(1, 0.02 * s0[key]), key
        assert np.abs(x0 - x1).max() < 1e-9 * np.abs(x0).max()


def test_schur_cg_as_one_launch_equals_the_launch_per_operation_solver():
    """NSX_CG_PERSISTENT=0 runs SolverCG on negative_S_tilde as separate launches (SpMV, dot, update, ILU, axpby): the
    persistent kernel k_cg_schur must walk through the same iteration history and end at the same vector."""
    p = Problem("cylinder", 3, 2, n_sub=24, ordering="colour")
    res = []
    for flag in ("0", "1"):
        os.environ["NSX_CG_PERSISTENT"] = flag
        try:
            dev, _ = _assembled(p)
            dev.prec_initialize(0)
            src = np.random.default_rng(3).standard_normal(p.dofs.n_dofs)
            y, st = dev.prec_vmult(0, src, inner_rtol=1e-10)
            t = dev.solve_time_step(0)  # reference tolerances: many short CG solves in a row (mailbox ring, region swap)
            res.append((st, y, t, dev.solution_owned.copy()))
            dev.close()
        finally:
            os.environ.pop("NSX_CG_PERSISTENT", None)
    (s0, y0, t0, x0), (s1, y1, t1, x1) = res
    assert s0["status"] == 0 and s1["status"] == 0
    assert s0["inner_S_iterations"] > 10 and abs(s0["inner_S_iterations"] - s1["inner_S_iterations"]) <= 1
    assert np.abs(y0 - y1).max() < 1e-8 * np.abs(y0).max()
    for key in ("outer_iterations", "inner_S_iterations"):
        assert abs(t0[key] - t1[key]) <= max(1, 0.05 * t0[key]), key
    assert np.abs(x0 - x1).max() < 1e-3 * np.abs(x0).max()


def test_schur_cg_persistent_kernel_is_the_one_that_runs():
    """the persistent launch shows up in the handle's kernel table (and the per-operation kernels of CG do not)"""
    p = Problem("cylinder", 3, 2, n_sub=24, ordering="colour")
    dev, _ = _assembled(p)
    dev.profile(True)
    dev.solve_time_step(0)
    table = dev.profile_table()
    dev.close()
    assert table.get("cg_S", {}).get("launches", 0) > 0
    assert table.get("cg_update", {}).get("launches", 0) == 0


def test_null_handle_and_unsorted_schur_blocks_are_argument_errors(prob):
    import ctypes as C
    import navierstokes_project_nm4pde_amd.nsx as nsx
    L = nsx.lib()
    buf = (C.c_double * 4)()
    for fn in (L.nsx_get_solution, L.nsx_get_solution_ghosted, L.nsx_get_rhs, L.nsx_set_rhs, L.nsx_set_solution):
        assert fn(None, buf) == -1
    assert L.nsx_set_ranks(None, 1, None, None) == -1
    dev = prob.device()
    ptr = np.asarray(prob.dofs.owned_p_ptr, dtype=np.int32).copy()
    ptr[1], ptr[2] = ptr[2], ptr[1]                 # first and last entry still right, the middle not ascending
    if ptr[1] != ptr[2]:
        with pytest.raises(nsx.NsxError) as e:
            dev.set_schur_blocks(ptr)
        assert e.value.code == -1
    dev.close()


def test_two_handles_driven_concurrently_on_one_device():
    """Two single-GPU handles alive on the same device, each driven by its own host thread.  The persistent kernels (Gram-Schmidt
    sweep, Schur CG) need all their workgroups resident; when the other handle's kernels hold compute units a grid may not be,
    the bounded waits then end the kernel without touching its vectors and the solve continues on the launch-per-operation
    path.  Either way both solves must end at the single-handle answer."""
    import threading
    p = Problem("cylinder", 3, 2, n_sub=24, ordering="colour")
    ref_dev, _ = _assembled(p)
    ref = ref_dev.solve_time_step(0, tol_abs=1e-10, inner_rtol=1e-8)
    x_ref = ref_dev.solution_owned.copy()
    ref_dev.close()
    devs = [_assembled(p)[0] for _ in range(2)]
    out, errs = [None, None], []

    def run(k):
        try:
            st = devs[k].solve_time_step(0, tol_abs=1e-10, inner_rtol=1e-8)
            out[k] = (st, devs[k].solution_owned.copy())
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for d in devs:
        d.close()
    assert not errs, errs
    for st, x in out:
        assert st["status"] == 0
        assert abs(st["outer_iterations"] - ref["outer_iterations"]) <= 2
        assert np.abs(x - x_ref).max() < 1e-7 * np.abs(x_ref).max()


def test_time_out_of_a_persistent_grid_falls_back_and_says_so():
    """NSX_GX_DROP_WG=k (fault injection): workgroup k of a persistent grid never posts its partial sums, which is what a grid
    that is not co-resident looks like.  The bounded waits must end the kernel without touching its vectors, the handle must
    redo the operation on the launch-per-operation path, stay there, leave its mailboxes clean, count the event
    (nsx_solve_stats::persistent_fallbacks, nsx_persistent_state) -- and end at the answer of a handle that never used the
    persistent kernels.  One Gram-Schmidt sweep and one Schur CG time out (2 s each)."""
    p = Problem("cylinder", 3, 2, n_sub=24, ordering="colour")
    res = {}
    for name, env in (("plain", {"NSX_MGS": "0", "NSX_CG_PERSISTENT": "0"}), ("healthy", {}), ("dropped", {"NSX_GX_DROP_WG": "1"})):
        os.environ.update(env)
        try:
            dev, _ = _assembled(p)
            st = dev.solve_time_step(0, tol_abs=1e-10, inner_rtol=1e-8)
            res[name] = (st, dev.solution_owned.copy(), dev.persistent_state())
            if name == "dropped":  # the handle stays usable, and stays on the fallback path
                st2 = dev.solve_time_step(0, tol_abs=1e-10, inner_rtol=1e-8)
                assert st2["status"] == 0 and st2["persistent_fallbacks"] == 2
                assert dev.persistent_state() == res[name][2]
            dev.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    (s0, x0, p0), (s1, x1, p1), (s2, x2, p2) = res["plain"], res["healthy"], res["dropped"]
    assert p0 == {"sweep_persistent": False, "cg_persistent": False, "fallbacks": 0, "dirty_mailbox_words": 0}
    assert p1 == {"sweep_persistent": True, "cg_persistent": True, "fallbacks": 0, "dirty_mailbox_words": 0}
    assert p2 == {"sweep_persistent": False, "cg_persistent": False, "fallbacks": 2, "dirty_mailbox_words": 0}
    assert s1["persistent_fallbacks"] == 0 and s2["persistent_fallbacks"] == 2
    for s, x in ((s1, x1), (s2, x2)):
        assert s["status"] == 0
        for key in ("outer_iterations", "inner_F_iterations", "inner_S_iterations"):
            assert abs(s0[key] - s[key]) <= max(1, 0.02 * s0[key]), key
        assert np.abs(x0 - x).max() < 1e-9 * np.abs(x0).max()
    # after the time-out the dropped handle runs the very kernels of the plain one
    assert np.abs(x0 - x2).max() < 1e-11 * np.abs(x0).max()


def test_schur_cg_with_the_operator_in_lds_is_bit_identical_to_the_slab_stream(capfd):
    """Blocks of at most 96 rows: k_cg_schur keeps the explicit block inverses in registers (NSX_CG_PRES) and the block's rows of
    negative_S_tilde in LDS (NSX_CG_LRES) for the whole solve.  Same lanes, same entries per lane, same order of every sum as the
    variant that streams the operator slabs in every iteration: the vectors are equal to the last bit.  The variant that streams the
    block inverses too sums each row of P_b g in another grouping: equal to rounding."""
    from navierstokes_project_nm4pde_amd.frontend import merge_ranks
    p = Problem("cylinder", 3, 3, n_sub=128, ordering="colour")
    assert np.diff(p.dofs.owned_p_ptr).max() <= 96
    res = {}
    os.environ["NSX_DEBUG"] = "1"
    try:
        for name, env in (("lds", {}), ("slabs", {"NSX_CG_LRES": "0"}), ("streamed", {"NSX_CG_PRES": "0"})):
            os.environ.update(env)
            try:
                dev = p.device()
                dev.set_schur_blocks(merge_ranks(p.dofs.owned_p_ptr, 96))
                dev.set_solution(p.smooth_velocity())
                import navierstokes_project_nm4pde_amd.nsx as nsx
                dev.assemble(nsx.TEMAM)
                dev.apply_boundary_values(*_bc(p, p.deltat))
                dev.prec_initialize(0)
                src = np.random.default_rng(8).standard_normal(p.dofs.n_dofs)
                y, st = dev.prec_vmult(0, src, inner_rtol=1e-10)
                t = dev.solve_time_step(0)
                res[name] = (y, st, t, dev.solution_owned.copy(), capfd.readouterr().err)
                dev.close()
            finally:
                for k in env:
                    os.environ.pop(k, None)
    finally:
        os.environ.pop("NSX_DEBUG", None)
    assert "block inverses in registers 1 (rows per lane group 6), operator in LDS 1" in res["lds"][4], res["lds"][4]
    assert "block inverses in registers 1 (rows per lane group 6), operator in LDS 0" in res["slabs"][4]
    assert "block inverses in registers 0" in res["streamed"][4]
    y0, s0, t0, x0, _ = res["lds"]
    assert s0["status"] == 0 and s0["inner_S_iterations"] > 10
    y1, s1, t1, x1, _ = res["slabs"]
    assert np.array_equal(y0, y1) and np.array_equal(x0, x1) and s0["inner_S_iterations"] == s1["inner_S_iterations"]
    assert t0["outer_iterations"] == t1["outer_iterations"] and t0["inner_S_iterations"] == t1["inner_S_iterations"]
    y2, s2, _, _, _ = res["streamed"]
    assert abs(s2["inner_S_iterations"] - s0["inner_S_iterations"]) <= 1
    assert np.abs(y0 - y2).max() < 1e-8 * np.abs(y0).max()


def test_ilu_factorisation_in_lds_is_bit_identical_to_the_one_through_global_memory():
    """k_ilu_factor_lds stages the in-block part of a whole rank block in LDS; NSX_ILU_FACTOR_LDS=0 is k_ilu_factor_small (dense
    row per wave, pivot rows read from global memory).  Same operations on the same entries in the same order: equal factors
    (F and the Schur matrix), equal packed stream (ilu_apply)."""
    import navierstokes_project_nm4pde_amd.nsx as nsx
    p = Problem("cylinder", 3, 2, n_sub=24, ordering="colour")
    res = []
    for flag in ("1", "0"):
        os.environ["NSX_ILU_FACTOR_LDS"] = flag
        try:
            dev = p.device()
            dev.set_solution(p.smooth_velocity())
            dev.assemble(nsx.TEMAM)
            dev.apply_boundary_values(*_bc(p, p.deltat))
            dev.prec_initialize(0)
            b = np.random.default_rng(2).standard_normal(p.dofs.n_u)
            res.append((dev.ilu(0)[2], dev.ilu(1)[2], dev.ilu_apply(0, b)))
            dev.close()
        finally:
            os.environ.pop("NSX_ILU_FACTOR_LDS", None)
    for x, y in zip(*res):
        assert np.abs(x).max() > 0 and np.array_equal(x, y)


def test_rank_tables_with_empty_ranks_factor_and_solve_like_the_table_without_them():
    """A rank that owns nothing (an MPI rank without cells on a coarse mesh) is an empty ILU block: the factorisation kernels, the
    packed triangular-solve stream and the Schur CG with its operator in LDS must step over it.  Same factors, same applications
    and the same time step as with the table that leaves the empty ranks out."""
    import navierstokes_project_nm4pde_amd.nsx as nsx
    p = Problem("cylinder", 3, 2, n_sub=24, ordering="colour")
    u_ptr, p_ptr = (np.asarray(a, dtype=np.int32) for a in (p.dofs.owned_u_ptr, p.dofs.owned_p_ptr))

    def with_empties(ptr):  # an empty rank in front, two in the middle, one behind
        return np.concatenate(([ptr[0]], ptr[:8], [ptr[7], ptr[7]], ptr[8:], [ptr[-1]])).astype(np.int32)

    res = []
    for up, pp in ((u_ptr, p_ptr), (with_empties(u_ptr), with_empties(p_ptr))):
        dev = p.device()
        dev.set_ranks(up, pp)
        dev.set_solution(p.smooth_velocity())
        dev.assemble(nsx.TEMAM)
        dev.apply_boundary_values(*_bc(p, p.deltat))
        dev.prec_initialize(0)
        rng = np.random.default_rng(4)
        zu = dev.ilu_apply(0, rng.standard_normal(p.dofs.n_u))
        zp = dev.ilu_apply(1, rng.standard_normal(p.dofs.n_p))
        st = dev.solve_time_step(0)
        res.append((dev.ilu(0)[2], dev.ilu(1)[2], zu, zp, dev.solution_owned.copy(), st))
        dev.close()
    a, b = res
    for k in range(4):
        assert np.abs(a[k]).max() > 0 and np.array_equal(a[k], b[k]), k
    # the time step: the chunks of the products and the waves of the triangular solves are cut along the rank table, so sums are
    # taken in another order: equal to the solver tolerance, same iteration history
    assert a[5]["status"] == 0 and b[5]["status"] == 0
    for key in ("outer_iterations", "inner_F_iterations", "inner_S_iterations"):
        assert abs(a[5][key] - b[5][key]) <= max(1, 0.05 * a[5][key]), key
    assert np.abs(a[4] - b[4]).max() < 1e-3 * np.abs(a[4]).max()


def test_lane_owner_solve_that_cannot_run_fails_the_call(monkeypatch):
    """k_ilu_solve_lanes needs its dynamic LDS array at address 0 (the stream holds absolute LDS addresses).  If that ever breaks,
    the kernel does not write x: the call must FAIL (mapped error word, ilu_check), not hand back stale memory.  The condition
    cannot be produced from outside, so the kernel's guard is forced (NSX_ILU_LDS_GUARD_TEST)."""
    from navierstokes_project_nm4pde_amd.nsx import NsxError
    p = Problem("cylinder", 3, 2, n_sub=24, ordering="colour")
    dev = p.device()
    dev.set_solution(p.smooth_velocity())
    dev.assemble(1)
    dev.prec_initialize(0)
    b = np.random.default_rng(2).standard_normal(p.dofs.n_u)
    good = dev.ilu_apply(0, b)
    assert np.isfinite(good).all()
    monkeypatch.setenv("NSX_ILU_LDS_GUARD_TEST", "1")
    with pytest.raises(NsxError) as e:
        dev.ilu_apply(0, b)
    assert "LDS" in str(e.value)
    monkeypatch.delenv("NSX_ILU_LDS_GUARD_TEST")
    assert (dev.ilu_apply(0, b) == good).all()   # the handle is usable again
    dev.close()


def test_norm_after_the_sweep_from_the_gram_matrix(prob):
    """|w'|^2 = |w|^2 - 2 h.r + h^T G h (k_mgs_one, k_ls_solve) against the explicitly summed norm, on the sweep kernel itself
    (nsx_gram_schmidt_cycle): (a) where the sweep leaves more than 1 % of the norm the two agree to 1e-13 relative and the
    default takes the formula; (b) a nearly dependent vector makes the default take the explicit sum (bitwise the always-explicit
    run), which is right to 1e-10 where the formula alone has lost most of its digits."""
    dev = prob.device()
    rng = np.random.default_rng(17)
    n, m = 200000, 12
    V = rng.standard_normal((m, n))
    V[5] = 0.3 * V[5] + V[:5].sum(axis=0)                  # a good share of it lies in the span of the vectors before it, > 1 % is left
    V[m - 1] = V[:m - 1].sum(axis=0) + 1e-7 * V[m - 1]     # nearly dependent: the sweep removes all but 1e-14 of |w|^2
    Qf, Hf, nf = dev.gram_schmidt_cycle(V, norm_guard=0.0)      # always the formula
    Qx, Hx, nx = dev.gram_schmidt_cycle(V, norm_guard=1e300)    # always the explicit sum
    Qd, Hd, nd = dev.gram_schmidt_cycle(V)                      # the library's threshold (1e-2)
    # reference: modified Gram-Schmidt in numpy
    Q = np.zeros_like(V)
    Q[0] = V[0] / np.linalg.norm(V[0])
    true_n2 = np.zeros(m)
    true_n2[0] = V[0] @ V[0]
    for k in range(1, m):
        w = V[k].copy()
        for i in range(k):
            w -= (w @ Q[i]) * Q[i]
        true_n2[k] = w @ w
        Q[k] = w / np.sqrt(true_n2[k])
    before = np.array([V[k] @ V[k] for k in range(m)])
    for k in range(1, m - 1):
        assert true_n2[k] > 0.01 * before[k]
        assert abs(nf[k] - nx[k]) <= 1e-13 * nx[k], (k, nf[k], nx[k])      # (a) formula == explicit sum
        assert nd[k] == nf[k]                                             # ... and it is the formula the default used
        assert abs(nx[k] - true_n2[k]) <= 1e-12 * true_n2[k]
    k = m - 1
    assert true_n2[k] < 1e-10 * before[k]
    # (b) the default refused the formula: it agrees with the always-explicit run to the rounding the EARLIER sweeps differ by
    # (those normalised their vectors with the formula's norm in one run and the explicit one in the other: 1e-13 apart)
    # (the vector itself is what is left of a cancellation by seven digits: the two runs' 1e-13 differences in the basis show up at 1e-7 in it)
    assert abs(nd[k] - nx[k]) <= 1e-9 * nx[k] and np.abs(Qd[k] - Qx[k]).max() <= 1e-5 * np.abs(Qx[k]).max()
    assert abs(nx[k] - true_n2[k]) <= 1e-8 * true_n2[k]
    assert abs(nf[k] - true_n2[k]) > 1e-6 * true_n2[k]                     # the formula alone: a difference of numbers 1e14 times larger
    assert np.abs(Hd - Hx).max() <= 1e-12 * np.abs(Hx).max()
    dev.close()


@pytest.mark.parametrize("cap,entries", [(23, 10), (19, 12)])
def test_larger_sweep_instantiations_equal_the_launch_per_link_chain(cap, entries):
    """k_mgs_one<10,8> and <12,6> (10 / 12 entries per thread, 8 / 6 basis vectors in registers) are what one GPU runs between 1.05 M
    and 1.57 M velocity entries; NSX_MGS_MAXWG caps the resident grid so that the 54 043-DoF mesh needs them (51 558 entries over
    23 x 256 threads: 9; over 19 x 256: 11).  Same history and solution as the chain of separate launches (NSX_MGS=0), and the
    handle says which instantiation ran."""
    p = Problem("cylinder", 3, 2, n_sub=8, ordering="colour")
    res = []
    for env in ({"NSX_MGS": "0"}, {"NSX_MGS_MAXWG": str(cap)}):
        os.environ.update(env)
        try:
            dev, _ = _assembled(p)
            st = dev.solve_time_step(3, tol_abs=1e-10, inner_rtol=1e-8)
            res.append((st, dev.solution_owned.copy(), dev.path_info(), dev.persistent_state()))
            dev.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    (s0, x0, i0, _), (s1, x1, i1, p1) = res
    assert i0["sweep_entries_per_thread_max"] == 0 and i1["sweep_entries_per_thread_max"] == entries, (i0, i1)
    assert i1["sweep_grid"] <= cap and p1["fallbacks"] == 0 and p1["sweep_persistent"]
    assert s0["status"] == 0 and s0["outer_iterations"] > 5
    for key in ("outer_iterations", "inner_F_iterations", "inner_S_iterations"):
        assert abs(s0[key] - s1[key]) <= max(1, 0.02 * s0[key]), key
    assert np.abs(x0 - x1).max() < 1e-9 * np.abs(x0).max()


@pytest.mark.parametrize("ranks", [600, 1100])
def test_schur_cg_in_two_launches_on_one_gpu_with_more_blocks_than_resident_workgroups(ranks):
    """More Schur ILU blocks than the persistent CG's resident grid holds (512): one GPU then runs the two launches per iteration of
    the distributed path (k_cgd_A / k_cgd_B); above 1024 blocks their per-block partial sums go through k_cgd_fold (the 10.6 M-DoF
    mesh on one GPU has 4 833 blocks).  Same CG iteration counts and solution as the launch-per-operation solver (NSX_CG_FUSED=0)."""
    import navierstokes_project_nm4pde_amd.nsx as nsx
    p = Problem("cylinder", 3, 3, ordering="first_touch")
    res = []
    for env in ({"NSX_CG_FUSED": "0"}, {}):
        os.environ.update(env)
        try:
            dev = p.device()
            dev.set_internal_layout(ranks, nsx.COLOUR, 0)
            dev.set_solution(p.smooth_velocity())
            dev.assemble(nsx.TEMAM)
            dev.apply_boundary_values(*_bc(p, p.deltat))
            dev.profile(True)
            dev.prec_initialize(0)
            src = np.random.default_rng(3).standard_normal(p.dofs.n_dofs)
            y, st = dev.prec_vmult(0, src, inner_rtol=1e-10)
            t = dev.solve_time_step(0)
            res.append((st, y, t, dev.solution_owned.copy(), dev.profile_table(), dev.path_info()))
            dev.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    (s0, y0, t0, x0, tab0, i0), (s1, y1, t1, x1, tab1, i1) = res
    assert i1["schur_blocks"] > 512 and (ranks < 1024 or i1["schur_blocks"] > 1024), i1
    assert i0["schur_cg_path"] == 1 and i1["schur_cg_path"] == 3, (i0, i1)
    assert tab1.get("cgd_A", {}).get("launches", 0) > 0 and tab1.get("cgd_B", {}).get("launches", 0) > 0
    assert tab1.get("spmv_S", {}).get("launches", 0) == 0 and tab0.get("spmv_S", {}).get("launches", 0) > 0
    assert s0["status"] == 0 and s1["status"] == 0
    assert s0["inner_S_iterations"] > 10 and abs(s0["inner_S_iterations"] - s1["inner_S_iterations"]) <= max(1, 0.02 * s0["inner_S_iterations"])
    assert np.abs(y0 - y1).max() < 1e-8 * np.abs(y0).max()
    for key in ("outer_iterations", "inner_S_iterations"):
        assert abs(t0[key] - t1[key]) <= max(1, 0.05 * t0[key]), key
    assert np.abs(x0 - x1).max() < 1e-3 * np.abs(x0).max()


@pytest.mark.parametrize("dim,level,n_sub", [(3, 2, 200), (3, 2, 160), (2, 3, 40)])
def test_triangular_solves_inside_the_sweeps_launch_equal_the_separate_kernels(dim, level, n_sub):
    """k_ilu_mgs: the two sweeps of the velocity ILU(0) solve (PreconditionILU::vmult inside every inner GMRES iteration on F) run in
    the launch of the Gram-Schmidt sweep that follows them -- workgroup b of the persistent grid is wave b of the solve's schedule, z
    goes from LDS straight into the sweep's registers.  z itself is bit for bit the separate kernel's; the sweep's sums are taken in
    another grouping, so the run walks through the same iteration history and ends at the same solution to rounding.
    NSX_ILU_MGS=1 switches it on; the default is the separate kernels (k_ilu_solve_lanes, k_mgs_one): at the bench size the fused launch
    is 4 % faster per outer iteration, but its re-rolled iteration history has more restart steps in both sampled windows."""
    p = Problem("cylinder", dim, level, n_sub=n_sub, ordering="colour")
    res = []
    for env in ({}, {"NSX_ILU_MGS": "1"}):
        os.environ.update(env)
        try:
            st, xs = [], []
            for k in (0, 3):   # Yosida (two F solves + CG), aSIMPLE (F + S by GMRES), each from the same fresh state
                dev, _ = _assembled(p)   # (a second solve from the converged state of the first has a right-hand side of rounding size: its iteration count is noise)
                dev.profile(True)
                st.append(dev.solve_time_step(k, tol_abs=1e-10, inner_rtol=1e-8))
                xs.append(dev.solution_owned.copy())
                last = (dev.path_info(), dev.persistent_state(), dev.profile_table())
                dev.close()
            res.append((st, np.concatenate(xs)) + last)
        finally:
            for k in env:
                os.environ.pop(k, None)
    (s0, x0, i0, p0, t0), (s1, x1, i1, p1, t1) = res
    assert i0["fused_launches"] == 0 and t0.get("ilu_mgs", {}).get("launches", 0) == 0 and t0.get("ilu_solve_F", {}).get("launches", 0) > 0
    assert i1["fused_launches"] > 20 and t1.get("ilu_mgs", {}).get("launches", 0) == i1["fused_launches"], (i1, t1.get("ilu_mgs"))
    # what is left of the separate kernel: the preconditioned residual at the start of every GMRES cycle
    assert t1.get("ilu_solve_F", {}).get("launches", 0) < 0.25 * t0["ilu_solve_F"]["launches"]
    assert p1["fallbacks"] == 0 and p1["sweep_persistent"] and p1["dirty_mailbox_words"] == 0
    for a, b in zip(s0, s1):
        assert a["status"] == 0 and b["status"] == 0
        for key in ("outer_iterations", "inner_F_iterations", "inner_S_iterations"):
            assert abs(a[key] - b[key]) <= max(1, 0.02 * a[key]), key
    assert np.abs(x0 - x1).max() < 1e-9 * np.abs(x0).max()


def test_block_inverses_built_in_lds_are_bit_identical_to_the_ones_built_through_global_memory():
    """k_ilu_invert_lds keeps a Schur ILU block's n x n inverse in LDS while a thread per column walks the rows (NSX_ILU_INVERT_LDS=0:
    k_ilu_invert, the same walk through global memory).  Same operations in the same order: the application of the factors
    (ilu_apply on the Schur matrix = one dense product per block) and a whole time step agree to the last bit."""
    import navierstokes_project_nm4pde_amd.nsx as nsx
    from navierstokes_project_nm4pde_amd.frontend import merge_ranks
    p = Problem("cylinder", 3, 3, n_sub=128, ordering="colour")
    res = []
    for flag in ("1", "0"):
        os.environ["NSX_ILU_INVERT_LDS"] = flag
        try:
            dev = p.device()
            dev.set_schur_blocks(merge_ranks(p.dofs.owned_p_ptr, 96))
            dev.set_solution(p.smooth_velocity())
            dev.assemble(nsx.TEMAM)
            dev.apply_boundary_values(*_bc(p, p.deltat))
            dev.prec_initialize(0)
            b = np.random.default_rng(2).standard_normal(p.dofs.n_p)
            z = dev.ilu_apply(1, b)
            st = dev.solve_time_step(0)
            res.append((z, dev.solution_owned.copy(), st, dev.schur().data.copy()))
            dev.close()
        finally:
            os.environ.pop("NSX_ILU_INVERT_LDS", None)
    (z0, x0, s0, v0), (z1, x1, s1, v1) = res
    assert np.abs(z0).max() > 0 and np.array_equal(z0, z1) and np.array_equal(x0, x1) and np.array_equal(v0, v1)
    assert s0["outer_iterations"] == s1["outer_iterations"] and s0["inner_S_iterations"] == s1["inner_S_iterations"]
