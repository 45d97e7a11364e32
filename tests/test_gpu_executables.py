"""GPU suite: the C++ host mirror executables and the convergence study, end to end through the C-ABI."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dim", [2, 3])
def test_host_mirror_executable_matches_python_driver(dim, tmp_path):
    import __graft_entry__ as ge
    ge.build()
    exe = os.path.join(ROOT, "navierstokes_project_nm4pde_amd", "host", "navier_stokes%dD" % dim)
    out = subprocess.run([exe, "level:1", "3", "4"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    its = [int(x) for x in re.findall(r"Result:\s+(\d+) GMRES iterations", out.stdout)]
    assert len(its) == 3
    # the same three steps driven from Python
    from navierstokes_project_nm4pde_amd import nsx
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    mesh = Mesh.cylinder(dim, 1).partition(1, 4)
    dofs, tables = DoFs(mesh), Tables(dim)
    dt = 2e-4 if dim == 3 else 0.01
    dev = nsx.Nsx(dofs, tables, 1e-3, dt)
    dev.set_solution(np.zeros(dofs.n_dofs))
    inlet = InletVelocity(dim, 2)
    t, py_its = 0.0, []
    for step in range(3):
        t += dt
        if step == 0:
            dev.assemble(nsx.TEMAM)
        else:
            dev.assemble_time_step(nsx.TEMAM if dim == 2 else 0)
        dev.apply_boundary_values(*cylinder_boundary_values(dofs, inlet, t))
        py_its.append(dev.solve_time_step(0 if dim == 3 else 3, inner_maxiter=100000 if dim == 3 else 10000)["outer_iterations"])
    assert its == py_its
    assert os.path.exists(tmp_path / ("timings_%dD.csv" % dim))


def test_convergence_study_on_device_matches_oracle():
    """Ethier-Steinmann known answer (reference main_convergence3D.cpp) on the GPU: same errors as the oracle."""
    import oracle
    from navierstokes_project_nm4pde_amd import nsx
    from navierstokes_project_nm4pde_amd.problem import run_convergence_case
    res_d = [run_convergence_case(lambda d, t, nu, dt: nsx.Nsx(d, t, nu, dt), n, tol_abs=1e-9, inner_rtol=1e-6) for n in (2, 4)]
    res_o = [run_convergence_case(lambda d, t, nu, dt: oracle.Oracle(d, t, nu, dt), n, tol_abs=1e-9, inner_rtol=1e-6) for n in (2, 4)]
    for rd, ro in zip(res_d, res_o):
        assert abs(rd["L2"] - ro["L2"]) < 1e-6 * ro["L2"] and abs(rd["H1"] - ro["H1"]) < 1e-6 * ro["H1"]
        assert np.abs(rd["solution"] - ro["solution"]).max() < 1e-6 * np.abs(ro["solution"]).max()
    assert np.log2(res_d[0]["L2"] / res_d[1]["L2"]) > 2.8
