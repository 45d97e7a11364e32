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


@pytest.mark.parametrize("dim,level,world,n_sub,prec,ordering,schur_merge",
                         [(3, 1, 2, 3, 0, "first_touch", 0), (2, 2, 3, 2, 3, "first_touch", 0),
                          (3, 1, 2, 4, 0, "colour", 2)])   # the bench's options: colour order, merged Schur blocks (dense inverses + fused dot)
def test_distributed_solve_equals_single_process(tmp_path, dim, level, world, n_sub, prec, ordering, schur_merge):
    out = tmp_path / "dist.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(ROOT, "tests", "dist_worker.py"), str(dim), str(level), str(n_sub), str(prec), str(out), ordering, str(schur_merge)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = np.load(out)
    # single-process reference with the same world * n_sub virtual ranks
    from navierstokes_project_nm4pde_amd import nsx
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    mesh = Mesh.cylinder(dim, level).partition(world, n_sub)
    dofs, tables = DoFs(mesh, ordering), Tables(dim)
    dt = 2e-4 if dim == 3 else 1e-2
    dev = nsx.Nsx(dofs, tables, 1e-3, dt)
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
        dev.apply_boundary_values(*cylinder_boundary_values(dofs, inlet, t))
        if step == 0:
            y = dev.system_vmult(d["x"])
            assert np.abs(y - d["vmult"]).max() < 1e-12 * np.abs(y).max()
        st = dev.solve_time_step(prec, tol_abs=1e-11, inner_rtol=1e-10)
        x = dev.solution_owned
        assert np.abs(x - d["sols"][step]).max() < 1e-8 * np.abs(x).max(), step
        assert abs(st["outer_iterations"] - int(d["iters"][step])) <= 1
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
        st = dev.solve_time_step(nsx.YOSIDA)
        out.append((st["outer_iterations"], st["inner_F_iterations"], dev.solution_owned))
        dev.close()
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert np.abs(out[0][2] - out[1][2]).max() < 1e-12 * np.abs(out[0][2]).max()
