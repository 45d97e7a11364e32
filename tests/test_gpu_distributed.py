"""GPU suite: the N > 1 path (owned + ghost rows, halo exchange, distributed dot products) on one GPU box:
`world` processes share cuda:0 and communicate through the host-callback backend over gloo; the result must equal
the single-process run with the same virtual ranks (the algorithm is partition-invariant by construction)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dim,level,world,n_sub,prec,ordering,schur_merge,layout_ranks",
                         [(3, 1, 2, 3, 0, "first_touch", 0, 0), (2, 2, 3, 2, 3, "first_touch", 0, 0),
                          (3, 1, 2, 4, 0, "colour", 2, 0),   # the options of rounds 1-3's bench: colour order, merged Schur blocks (dense inverses)
                          (3, 2, 2, 1, 0, "first_touch", 0, 12),   # round 4's bench: deal.II's numbering, one rank per GPU, the layout built inside every handle
                          (3, 2, 4, 1, 0, "first_touch", 0, 6)])   # the same with four ranks: several neighbours per rank, 6 / 9 / 12 / 15 virtual ranks inside them
def test_distributed_solve_equals_single_process(tmp_path, dim, level, world, n_sub, prec, ordering, schur_merge, layout_ranks):
    out = tmp_path / "dist.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(ROOT, "tests", "dist_worker.py"), str(dim), str(level), str(n_sub), str(prec), str(out), ordering, str(schur_merge),
           str(layout_ranks)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = np.load(out)
    if prec == 0:  # Yosida: the Schur CG of a distributed run is the two-launch one (nsx_cg.hip), not the launch-per-operation solver
        assert "cgd_A" in d["scopes"] and "cgd_B" in d["scopes"] and "spmv_S" not in d["scopes"], list(d["scopes"])
    # every F->vmult of a distributed handle goes through the LDS-staged SpMV: the chunks without a ghost column while the exchange is in
    # flight ("spmv_F"), the others behind it ("spmv_F_if"), the wait for the ghosts a scope of its own
    info = dict(zip(d["path_keys"], d["path_info"]))
    assert info["spmv_lds_staged"] == 1 and 0 < info["spmv_chunks_behind_halo"] <= info["spmv_chunks"], info
    assert "spmv_F" in d["scopes"] and "spmv_F_if" in d["scopes"] and "halo_u_wait" in d["scopes"], list(d["scopes"])
    assert info["neighbours"] >= 1 and info["ghost_nodes"] > 0
    # single-process reference with the same world * n_sub virtual ranks
    from navierstokes_project_nm4pde_amd import nsx
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values, obstacle_faces
    mesh = Mesh.cylinder(dim, level).partition(world, n_sub)
    dofs, tables = DoFs(mesh, ordering), Tables(dim)
    dt = 2e-4 if dim == 3 else 1e-2
    dev = nsx.Nsx(dofs, tables, 1e-3, dt)
    if layout_ranks:  # the same request on ONE handle that holds both ranks of the caller: the virtual ranks are dealt to the ranks in
        dev.set_internal_layout(world * layout_ranks, nsx.COLOUR, 24)  # proportion to their nodes, so the blocks may differ from the workers'
    dev.set_force_faces(*obstacle_faces(mesh), Tables(dim, Tables.FACE))
    if schur_merge:
        dev.set_schur_blocks(np.ascontiguousarray(dofs.owned_p_ptr[::schur_merge]))
    dev.set_solution(d["u0"])
    inlet = InletVelocity(dim, 2 if dim == 3 else 3)
    t = 0.0
    for step in range(3):
        t += dt
        if step == 0:
            dev.assemble(nsx.TEMAM)
        else:
            dev.assemble_time_step(nsx.TEMAM if dim == 2 else 0)
        bd, bv = cylinder_boundary_values(dofs, inlet, t)
        dev.apply_boundary_values(bd, bv)
        if step == 0:
            y = dev.system_vmult(d["x"])
            free = np.ones(len(y), dtype=bool)
            if layout_ranks:  # the two runs cut their virtual ranks differently, and a constrained row carries its RANK's diagonal value
                free[bd] = False   # (apply_boundary_values: first non-zero diagonal entry of the rank's range): those rows differ by construction
            assert np.abs(y - d["vmult"])[free].max() < 1e-12 * np.abs(y).max()
        st = dev.solve_time_step(prec, tol_abs=1e-10, inner_rtol=1e-10)  # 1e-11 sits on the floor the inner solves leave: 18 or 30 iterations by rounding
        x = dev.solution_owned
        assert np.abs(x - d["sols"][step]).max() < 1e-8 * np.abs(x).max(), step
        # (with the internal layout the two runs may cut the virtual ranks differently: another block-Jacobi ILU, the same solution)
        assert abs(st["outer_iterations"] - int(d["iters"][step])) <= (1 if not layout_ranks else max(3, 0.3 * st["outer_iterations"]))
        # distributed compute_forces (per-rank face integrals + the 2-double all-reduce of NavierStokes3D.cpp:830-831)
        f1, fw = np.array(dev.compute_forces()), d["forces"][step]
        assert np.abs(f1 - fw).max() < 1e-7 * max(1e-30, np.abs(f1).max()), (step, f1, fw)
    dev.close()


def test_rccl_single_rank_communicator_matches_plain_solve():
    """The RCCL calls themselves (ncclCommInitRank, ncclAllReduce on the compute stream after every reduction) on a
    1-rank communicator: same iteration counts and solution as without a communicator."""
    from navierstokes_project_nm4pde_amd import nsx
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    mesh = Mesh.cylinder(3, 1).partition(1, 4)
    dofs, tables = DoFs(mesh), Tables(3)
    out = []
    for use_comm in (False, True):
        dev = nsx.Nsx(dofs, tables, 1e-3, 2e-4)
        if use_comm:
            dev.comm_init_single()
        dev.set_solution(np.zeros(dofs.n_dofs))
        dev.assemble(nsx.TEMAM)
        dev.apply_boundary_values(*cylinder_boundary_values(dofs, InletVelocity(3), 2e-4))
        dev.profile(True)
        st = dev.solve_time_step(nsx.YOSIDA)
        out.append((st["outer_iterations"], st["inner_F_iterations"], dev.solution_owned, dev.profile_table(), st["inner_S_iterations"]))
        dev.close()
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert np.abs(out[0][2] - out[1][2]).max() < 1e-12 * np.abs(out[0][2]).max()
    # which Schur CG ran: the persistent launch without a communicator, the two launches per iteration with one (nsx_cg.hip)
    t0, t1 = out[0][3], out[1][3]
    assert t0.get("cg_S", {}).get("launches", 0) > 0 and t0.get("cgd_A", {}).get("launches", 0) == 0
    assert t1.get("cgd_A", {}).get("launches", 0) > 0 and t1.get("cgd_B", {}).get("launches", 0) > 0 and t1.get("cg_S", {}).get("launches", 0) == 0
    assert t1.get("spmv_S", {}).get("launches", 0) == 0                     # ... and not the launch-per-operation solver
    assert abs(out[0][4] - out[1][4]) <= max(1, 0.02 * out[0][4])            # same CG iteration count (sums in another fixed order)


def test_one_collective_per_gram_schmidt_sweep_equals_the_chain():
    """Distributed orthogonalisation: mgs_lowsync (all dots of a sweep, the new row of the basis' Gram matrix and |w|^2 in ONE
    all-reduce; |w|^2 after the sweep follows from the same numbers: NSX_MGS_LOWSYNC=2, the default; =1 pays a second collective
    for it) against NSX_MGS_LOWSYNC=0 (one all-reduce per link of SolverGMRES' add_and_dot chain, what the reference's MPI run
    pays), all through a 1-rank RCCL communicator so that the collectives really run.  Same iteration history, same solution (the
    coefficients are the chain's by linearity of the dot product), and about one collective per Krylov vector."""
    from navierstokes_project_nm4pde_amd import nsx
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    mesh = Mesh.cylinder(3, 2).partition(1, 8)
    dofs, tables = DoFs(mesh, "colour"), Tables(3)
    out = []
    for flag in ("0", "1", "2"):
        os.environ["NSX_MGS_LOWSYNC"] = flag
        os.environ["NSX_MGS_DIST"] = "0"   # the two-pass sweep itself (the default puts the collective inside the persistent grid: next test)
        try:
            dev = nsx.Nsx(dofs, tables, 1e-3, 2e-4)
            dev.comm_init_single()
            dev.set_solution(np.zeros(dofs.n_dofs))
            dev.assemble(nsx.TEMAM)
            dev.apply_boundary_values(*cylinder_boundary_values(dofs, InletVelocity(3), 2e-4))
            dev.profile(True)
            st = dev.solve_time_step(nsx.ASIMPLE, tol_abs=1e-10, inner_rtol=1e-8)   # aSIMPLE: every inner solve is a GMRES (Prec.hpp:271-289)
            out.append((st, dev.solution_owned.copy(), dev.comm_counters(), dev.profile_table()))
            dev.close()
        finally:
            os.environ.pop("NSX_MGS_LOWSYNC", None)
            os.environ.pop("NSX_MGS_DIST", None)
    (s0, x0, c0, t0), (s1, x1, c1, t1), (s2, x2, c2, t2) = out
    for s, x in ((s1, x1), (s2, x2)):
        assert s0["status"] == 0 and s["status"] == 0
        for key in ("outer_iterations", "inner_F_iterations", "inner_S_iterations"):
            assert abs(s0[key] - s[key]) <= max(1, 0.02 * s0[key]), key
        assert np.abs(x0 - x).max() < 1e-9 * np.abs(x0).max()
    assert t1.get("mgs_dots", {}).get("launches", 0) > 0 and t0.get("mgs_dots", {}).get("launches", 0) == 0
    # all-reduces per Krylov vector (every vector of this solve is a GMRES vector; each solve adds its two residual norms):
    # the chain about (dim + 1), the two-collective sweep 2, the default 1
    vectors = s2["outer_iterations"] + s2["inner_F_iterations"] + s2["inner_S_iterations"]
    assert c1[0] < 0.5 * c0[0], (c0, c1)
    assert c2[0] < 1.2 * vectors and c2[0] < 0.62 * c1[0], (c2, c1, vectors)


def test_collective_inside_the_persistent_sweep_equals_the_two_pass_sweep():
    """Distributed orthogonalisation, default path: the persistent one-exchange sweep with the all-reduce INSIDE its grid exchange
    (k_mgs_one<.., true>: reducers -> k_ext_wait -> ncclAllReduce -> k_ext_release on the communication stream, the grid waits for
    the flag with the ten newest basis vectors still in registers) against the two-pass sweep (NSX_MGS_DIST=0) and against a handle
    without a communicator, through a 1-rank RCCL communicator.  Same iteration history, same solution, the same number of
    collectives as the two-pass sweep (about one per Krylov vector), no time-out."""
    from navierstokes_project_nm4pde_amd import nsx
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    mesh = Mesh.cylinder(3, 2).partition(1, 8)
    dofs, tables = DoFs(mesh, "colour"), Tables(3)
    out = []
    for comm, dist in ((False, None), (True, "0"), (True, "1")):
        if dist is not None:
            os.environ["NSX_MGS_DIST"] = dist
        try:
            dev = nsx.Nsx(dofs, tables, 1e-3, 2e-4)
            if comm:
                dev.comm_init_single()
            dev.set_solution(np.zeros(dofs.n_dofs))
            dev.assemble(nsx.TEMAM)
            dev.apply_boundary_values(*cylinder_boundary_values(dofs, InletVelocity(3), 2e-4))
            dev.profile(True)
            st = dev.solve_time_step(nsx.ASIMPLE, tol_abs=1e-10, inner_rtol=1e-8)   # aSIMPLE: every inner solve is a GMRES
            out.append((st, dev.solution_owned.copy(), dev.comm_counters(), dev.profile_table(), dev.persistent_state()))
            dev.close()
        finally:
            os.environ.pop("NSX_MGS_DIST", None)
    (s0, x0, c0, t0, p0), (s1, x1, c1, t1, p1), (s2, x2, c2, t2, p2) = out
    assert s0["status"] == 0 and s1["status"] == 0 and s2["status"] == 0
    for s, x in ((s1, x1), (s2, x2)):
        for key in ("outer_iterations", "inner_F_iterations", "inner_S_iterations"):
            assert abs(s0[key] - s[key]) <= max(1, 0.02 * s0[key]), key
        assert np.abs(x0 - x).max() < 1e-9 * np.abs(x0).max()
    # which kernels ran: two passes without the switch, ONE persistent launch per sweep with it
    assert t1.get("mgs_dots", {}).get("launches", 0) > 0 and t1.get("mgs_sweep", {}).get("launches", 0) == 0
    assert t2.get("mgs_sweep", {}).get("launches", 0) > 0 and t2.get("mgs_dots", {}).get("launches", 0) == 0
    assert p2["fallbacks"] == 0 and p2["sweep_persistent"] and p2["dirty_mailbox_words"] == 0
    # the collective count is the two-pass sweep's (+ the one agreement of the ranks on the path, once per handle)
    assert abs(c2[0] - c1[0]) <= 2 + 0.01 * c1[0], (c1, c2)


def test_time_out_of_the_sweep_with_the_collective_inside_falls_back_on_every_rank():
    """The safety net of the default distributed sweep (NSX_GX_DROP_WG=1: one workgroup of the persistent grid never posts its sums,
    which is what a grid that is not co-resident -- e.g. with RCCL's own kernel -- looks like): the reducers' count stays incomplete,
    k_ext_wait raises the failure word, the all-reduce carries it to every rank, the grid ends without writing w, the handle
    switches to the two-pass sweep for good and SAYS so.  Same iteration history and solution as the two-pass sweep."""
    from navierstokes_project_nm4pde_amd import nsx
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    mesh = Mesh.cylinder(3, 2).partition(1, 8)
    dofs, tables = DoFs(mesh, "colour"), Tables(3)
    out = []
    for env in ({"NSX_MGS_DIST": "0"}, {"NSX_GX_DROP_WG": "1", "NSX_CG_PERSISTENT": "0"}):
        os.environ.update(env)
        try:
            dev = nsx.Nsx(dofs, tables, 1e-3, 2e-4)
            dev.comm_init_single()
            dev.set_solution(np.zeros(dofs.n_dofs))
            dev.assemble(nsx.TEMAM)
            dev.apply_boundary_values(*cylinder_boundary_values(dofs, InletVelocity(3), 2e-4))
            dev.profile(True)
            st = dev.solve_time_step(nsx.ASIMPLE, tol_abs=1e-10, inner_rtol=1e-8)
            out.append((st, dev.solution_owned.copy(), dev.profile_table(), dev.persistent_state()))
            dev.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    (s0, x0, t0, p0), (s1, x1, t1, p1) = out
    assert s0["status"] == 0 and s1["status"] == 0
    for key in ("outer_iterations", "inner_F_iterations", "inner_S_iterations"):
        assert abs(s0[key] - s1[key]) <= max(1, 0.02 * s0[key]), key
    assert np.abs(x0 - x1).max() < 1e-9 * np.abs(x0).max()
    assert p1["fallbacks"] >= 1 and not p1["sweep_persistent"]
    assert t1.get("mgs_sweep", {}).get("launches", 0) >= 1 and t1.get("mgs_dots", {}).get("launches", 0) > 0   # one failed launch, two passes from then on


def test_masked_streams_keep_the_collective_inside_the_larger_grids():
    """Above 917 k entries per GPU the sweep needs the 10- / 12-entry instantiations, which leave RCCL's kernel (264 VGPRs) no room on
    any CU they touch: the compute stream is then replaced by one whose CU mask leaves one CU per XCD out, and the communication
    stream confined to those eight (comm_reserve_cus).  NSX_COMM_CU_RESERVE=2 forces that on a small mesh, NSX_EXT_SELF_P2P=1 puts a
    real RCCL kernel (a self-addressed send / receive) in front of every collective of the sweep: persistent sweep, no time-out, the
    two-pass sweep's history and solution."""
    from navierstokes_project_nm4pde_amd import nsx
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    mesh = Mesh.cylinder(3, 2).partition(1, 8)
    dofs, tables = DoFs(mesh, "colour"), Tables(3)
    out = []
    for env in ({"NSX_MGS_DIST": "0"}, {"NSX_COMM_CU_RESERVE": "2", "NSX_EXT_SELF_P2P": "1"}):
        os.environ.update(env)
        try:
            dev = nsx.Nsx(dofs, tables, 1e-3, 2e-4)
            dev.comm_init_single()
            dev.set_solution(np.zeros(dofs.n_dofs))
            dev.assemble(nsx.TEMAM)
            dev.apply_boundary_values(*cylinder_boundary_values(dofs, InletVelocity(3), 2e-4))
            dev.profile(True)
            st = dev.solve_time_step(nsx.ASIMPLE, tol_abs=1e-10, inner_rtol=1e-8)
            out.append((st, dev.solution_owned.copy(), dev.profile_table(), dev.persistent_state()))
            dev.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    (s0, x0, t0, p0), (s1, x1, t1, p1) = out
    assert s0["status"] == 0 and s1["status"] == 0
    for key in ("outer_iterations", "inner_F_iterations", "inner_S_iterations"):
        assert abs(s0[key] - s1[key]) <= max(1, 0.02 * s0[key]), key
    assert np.abs(x0 - x1).max() < 1e-9 * np.abs(x0).max()
    assert p1["fallbacks"] == 0 and p1["sweep_persistent"] and p1["dirty_mailbox_words"] == 0
    assert t1.get("mgs_sweep", {}).get("launches", 0) > 0 and t1.get("mgs_dots", {}).get("launches", 0) == 0


@pytest.mark.parametrize("cap,entries", [(23, 10), (19, 12)])
def test_masks_switch_on_by_themselves_for_the_larger_distributed_grids(cap, entries):
    """k_mgs_one<10,8,true> / <12,6,true>: what the 10.6 M-DoF mesh on 8 GPUs runs (11.1 entries per thread).  NSX_MGS_MAXWG caps the
    grids so that the 54 043-DoF mesh needs 9-10 / 11-12 entries per thread: the 8-entry grid does not hold the vectors, so with the
    DEFAULT NSX_COMM_CU_RESERVE the ranks agree to mask their streams (compute: all CUs but one per XCD; communication: those eight)
    and the larger instantiation keeps the collective inside, with a real RCCL kernel (self-addressed send / receive) in front of it.
    Persistent, no fall-back, the two-pass sweep's history and solution."""
    from navierstokes_project_nm4pde_amd import nsx
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    mesh = Mesh.cylinder(3, 2).partition(1, 8)
    dofs, tables = DoFs(mesh, "colour"), Tables(3)
    out = []
    for env in ({"NSX_MGS_DIST": "0"}, {"NSX_MGS_MAXWG": str(cap), "NSX_EXT_SELF_P2P": "1"}):
        os.environ.update(env)
        try:
            dev = nsx.Nsx(dofs, tables, 1e-3, 2e-4)
            dev.comm_init_single()
            dev.set_solution(np.zeros(dofs.n_dofs))
            dev.assemble(nsx.TEMAM)
            dev.apply_boundary_values(*cylinder_boundary_values(dofs, InletVelocity(3), 2e-4))
            dev.profile(True)
            st = dev.solve_time_step(nsx.ASIMPLE, tol_abs=1e-10, inner_rtol=1e-8)
            out.append((st, dev.solution_owned.copy(), dev.profile_table(), dev.persistent_state(), dev.path_info()))
            if "NSX_MGS_MAXWG" in env:
                # a second communicator on the same handle: the masked stream belonged to the first one (the handle is back on all CUs),
                # the new one asks for it again and the sweep stays persistent
                dev.comm_init_single()
                assert dev.path_info()["cus_reserved"] == 0
                dev.set_solution(np.zeros(dofs.n_dofs))
                st2 = dev.solve_time_step(nsx.ASIMPLE, tol_abs=1e-10, inner_rtol=1e-8)
                i2, p2 = dev.path_info(), dev.persistent_state()
                assert st2["status"] == 0 and abs(st2["outer_iterations"] - st["outer_iterations"]) <= 1
                assert i2["cus_reserved"] == 8 and i2["sweep_collective_inside"] == 1 and p2["fallbacks"] == 0 and p2["sweep_persistent"]
            dev.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    (s0, x0, t0, p0, i0), (s1, x1, t1, p1, i1) = out
    assert i1["cus_reserved"] == 8 and i1["sweep_entries_per_thread_max"] == entries and i1["sweep_collective_inside"] == 1, i1
    assert s0["status"] == 0 and s1["status"] == 0
    for key in ("outer_iterations", "inner_F_iterations", "inner_S_iterations"):
        assert abs(s0[key] - s1[key]) <= max(1, 0.02 * s0[key]), key
    assert np.abs(x0 - x1).max() < 1e-9 * np.abs(x0).max()
    assert p1["fallbacks"] == 0 and p1["sweep_persistent"] and p1["dirty_mailbox_words"] == 0
    assert t1.get("mgs_sweep", {}).get("launches", 0) > 0 and t1.get("mgs_dots", {}).get("launches", 0) == 0


def test_a_collective_that_comes_too_late_for_one_ranks_grid_keeps_the_ranks_in_step():
    """NSX_EXT_LATE_RELEASE=k: the k-th collective inside a sweep never tells its grid that it is complete -- to ONE rank that is a
    collective arriving after the bounded wait (8 s), while its peers' grids may have gone on.  The rank must not change its collective
    sequence on its own: it finishes that sweep from the sums the late collective delivered (no further collective), asks all ranks
    through the next sweep's collective to leave the persistent path, and everybody continues on the two-pass sweep.  Two events are
    counted, the history and the solution are the two-pass sweep's, and a second communicator on the same handle starts afresh."""
    from navierstokes_project_nm4pde_amd import nsx
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    mesh = Mesh.cylinder(3, 1).partition(1, 4)
    dofs, tables = DoFs(mesh, "colour"), Tables(3)
    out = []
    for env in ({"NSX_MGS_DIST": "0"}, {"NSX_EXT_LATE_RELEASE": "7"}):
        os.environ.update(env)
        try:
            dev = nsx.Nsx(dofs, tables, 1e-3, 2e-4)
            dev.comm_init_single()
            dev.set_solution(np.zeros(dofs.n_dofs))
            dev.assemble(nsx.TEMAM)
            dev.apply_boundary_values(*cylinder_boundary_values(dofs, InletVelocity(3), 2e-4))
            dev.profile(True)
            st = dev.solve_time_step(nsx.ASIMPLE, tol_abs=1e-10, inner_rtol=1e-8)
            rec = [st, dev.solution_owned.copy(), dev.profile_table(), dev.persistent_state(), dev.comm_counters()]
            if "NSX_EXT_LATE_RELEASE" in env:   # a new communicator on the same handle: the sweep's path is chosen again
                dev.comm_init_single()
                st2 = dev.solve_time_step(nsx.ASIMPLE, tol_abs=1e-10, inner_rtol=1e-8)
                rec.append(st2)
            out.append(rec)
            dev.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    (s0, x0, t0, p0, c0), (s1, x1, t1, p1, c1, s2) = out
    assert s0["status"] == 0 and s1["status"] == 0 and s2["status"] == 0
    for key in ("outer_iterations", "inner_F_iterations", "inner_S_iterations"):
        assert abs(s0[key] - s1[key]) <= max(1, 0.02 * s0[key]), key
    assert np.abs(x0 - x1).max() < 1e-9 * np.abs(x0).max()
    assert p1["fallbacks"] == 2 and not p1["sweep_persistent"]       # the late collective, then the agreed exit
    assert t1.get("mgs_sweep", {}).get("launches", 0) == 8 and t1.get("mgs_dots", {}).get("launches", 0) > 0
    # no collective was added or lost on the way: 7 sweeps with one collective each, the leave sweep (one inside + its two-pass redo), two passes after
    assert abs(c1[0] - c0[0]) <= 12, (c0, c1)   # (+ the ranks' agreements on the path: once per handle and per role of a vector)


def test_rccl_ghost_exchange_branch_runs_with_the_rank_as_its_own_neighbour():
    """comm_halo_begin / comm_halo_finish over RCCL: pack kernel on the communication stream, grouped ncclSend / ncclRecv straight into
    the ghost region, event, wait of the compute stream.  RCCL refuses two ranks on one device, so the multi-process tests of this
    file exchange through host callbacks; here the RCCL branch itself runs, on a 1-rank communicator whose only neighbour is the
    rank itself (a real launch of RCCL's point-to-point kernel), for 1, 2 and 3 values per node and three exchanges in a row."""
    from navierstokes_project_nm4pde_amd import nsx
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables
    mesh = Mesh.cylinder(3, 1).partition(1, 4)
    dev = nsx.Nsx(DoFs(mesh), Tables(3), 1e-3, 2e-4)
    dev.comm_init_single()
    h0 = dev.comm_counters()[1]
    for n_own, n_ghost, ncomp in ((1000, 37, 1), (5000, 4999, 3), (120000, 30000, 3), (777, 2000, 2)):
        assert dev.comm_self_halo_test(n_own, n_ghost, ncomp) == 0.0, (n_own, n_ghost, ncomp)
    assert dev.comm_counters()[1] - h0 == 12
    dev.close()
