#!/usr/bin/env python3
"""BASELINE.json configs[4] at the size the CPU oracle can follow: Re = 100 (u_m = 2.25, NavierStokes3D.hpp:37,80), 500 consecutive
time steps of NavierStokes::solve() (NavierStokes3D.cpp:687-741: assemble / assemble_time_step, Dirichlet values, solve_time_step,
compute_forces) on the device and on the oracle side by side, tight tolerances (1e-12 abs / 1e-10 inner) so that the two runs can
be compared step by step; prints the largest deviation of c_D, c_L and of the solution over the series.

    python tools/drag_lift_series.py [--steps 500] [--level 1] > profiles/rNN_drag_lift_series.txt      (on an MI355X)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--n-sub", type=int, default=6)
    args = ap.parse_args()
    import navierstokes_project_nm4pde_amd.nsx as nsx
    from conftest import Problem
    from navierstokes_project_nm4pde_amd.frontend import Tables
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values, force_coefficients, obstacle_faces
    p = Problem("cylinder", 3, args.level, n_sub=args.n_sub, ordering="colour")
    dev, ora = p.device(), p.oracle()
    inlet = InletVelocity(3, test_case=2, u_m=2.25)
    cells, lf = obstacle_faces(p.mesh)
    ftab = Tables(3, Tables.FACE)
    dev.set_force_faces(cells, lf, ftab)
    dev.set_solution(np.zeros(p.dofs.n_dofs))
    print("# 3D cylinder level %d: %d DoF, %d virtual ranks, dt = %g, u_m = 2.25 (Re = 100), Yosida, tol 1e-12 / inner 1e-10"
          % (args.level, p.dofs.n_dofs, args.n_sub, p.deltat))
    print("# step  time  c_D(device)  c_L(device)  |c_D - oracle|  |c_L - oracle|  max|x - oracle| / max|oracle|  outer(dev) outer(oracle)")
    t, worst = 0.0, [0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    for step in range(args.steps):
        t += p.deltat
        bd, bv = cylinder_boundary_values(p.dofs, inlet, t)
        for o in (dev, ora):
            if step == 0:
                o.assemble(nsx.TEMAM)
            else:
                o.assemble_time_step(0)
            o.apply_boundary_values(bd, bv)
        sd = dev.solve_time_step(nsx.YOSIDA, tol_abs=1e-12, inner_rtol=1e-10)
        so = ora.solve_time_step(nsx.YOSIDA, tol_abs=1e-12, inner_rtol=1e-10)
        cd = force_coefficients(3, *dev.compute_forces(), mean_v=inlet.mean_velocity())
        co = force_coefficients(3, *ora.compute_forces(cells, lf, ftab), mean_v=inlet.mean_velocity())
        ex = np.abs(dev.solution_owned - ora.solution_owned).max() / np.abs(ora.solution_owned).max()
        e = (abs(cd[0] - co[0]), abs(cd[1] - co[1]), ex)
        worst = [max(a, b) for a, b in zip(worst, e)]
        if step < 5 or step % 25 == 24 or step == args.steps - 1:
            print("%4d  %.4f  %+.10e  %+.10e  %.2e  %.2e  %.2e  %d %d" % (step + 1, t, cd[0], cd[1], e[0], e[1], e[2], sd["outer_iterations"], so["outer_iterations"]), flush=True)
    print("# %d steps in %.0f s: max |c_D - oracle| = %.2e, max |c_L - oracle| = %.2e, max relative solution difference = %.2e"
          % (args.steps, time.perf_counter() - t0, worst[0], worst[1], worst[2]))
    dev.close()
    ok = worst[0] < 1e-8 and worst[1] < 1e-8 and worst[2] < 1e-8
    print("# north_star bound 1e-8 on drag / lift: %s" % ("met" if ok else "NOT met"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
