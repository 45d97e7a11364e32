#!/usr/bin/env python3
"""External sanity run (SURVEY 8c pin 6; the reference's own check, NavierStokes2D.cpp:746-748,855): the DFG benchmark 2D-3 of
Schaefer & Turek (1996) through the device path -- channel 2.2 x 0.41 with the cylinder of mesh/Cylinder2D.geo, nu = 1e-3,
inflow 4 u_m y (H - y) sin(pi t / 8) / H^2 with u_m = 1.5 (test_case 2, NavierStokes2D.hpp:33-34), T = 8, the reference's
defaults deltat = 1e-2 and aSIMPLE (main2D.cpp:21-22, NavierStokes2D.cpp:547), Temam term kept in the time loop
(NavierStokes2D.cpp:446), coefficients normalised with the CONSTANT mean velocity 2 u_m / 3 = 1 that getMeanVelocity() returns
for test case 2 (NavierStokes2D.hpp:66-74) -- which is the benchmark's own normalisation.

Published reference intervals (Schaefer & Turek 1996, table for 2D-3): c_D max 2.93 - 2.97, c_L max 0.47 - 0.49,
Delta p (t = 8 s) -0.115 ... -0.105.  This pins nothing formally -- the reference holds no such numbers -- but it is the one check
not written from the same memory as the oracle: boundary rule, force formulas and the time loop have to be right for it.

    python tools/schaefer_turek_2d.py [--levels 4 8] [--deltat 0.01] > profiles/rNN_schaefer_turek_2d.txt      (on an MI355X)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(level, deltat, T, n_sub):
    import navierstokes_project_nm4pde_amd.nsx as nsx
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables
    from navierstokes_project_nm4pde_amd.problem import (InletVelocity, cylinder_boundary_values, force_coefficients, obstacle_faces,
                                                         pressure_difference)
    mesh = Mesh.cylinder(2, level).partition(1, n_sub)
    dofs, tables = DoFs(mesh, "colour"), Tables(2)
    dev = nsx.Nsx(dofs, tables, 1e-3, deltat)
    inlet = InletVelocity(2, test_case=2, u_m=1.5)
    dev.set_force_faces(*obstacle_faces(mesh), Tables(2, Tables.FACE))
    dev.set_solution(np.zeros(dofs.n_dofs))
    n_steps = int(round(T / deltat))
    t, cds, cls, its = 0.0, [], [], []
    t0 = time.perf_counter()
    for step in range(n_steps):
        t += deltat
        if step == 0:
            dev.assemble(nsx.TEMAM)
        else:
            dev.assemble_time_step(nsx.TEMAM)
        dev.apply_boundary_values(*cylinder_boundary_values(dofs, inlet, t))
        st = dev.solve_time_step(nsx.ASIMPLE, inner_maxiter=10000)   # reference tolerances: 1e-4 abs, inner 1e-2
        its.append(st["outer_iterations"])
        cd, cl = force_coefficients(2, *dev.compute_forces(), mean_v=inlet.mean_velocity())
        cds.append(cd)
        cls.append(cl)
        if step % 100 == 99:  # progress (a silent GPU job is taken to be hung after 7 minutes)
            print("level %d: step %d / %d, c_D %.4f c_L %.4f" % (level, step + 1, n_steps, cd, cl), file=sys.stderr, flush=True)
    dp = pressure_difference(mesh, dofs, dev.solution)   # p(0.15, 0.2) - p(0.25, 0.2) at t = T
    wall = time.perf_counter() - t0
    dev.close()
    cds, cls = np.array(cds), np.array(cls)
    return {"level": level, "cells": int(mesh.cells.shape[0]), "n_dofs": int(dofs.n_dofs), "steps": n_steps, "deltat": deltat,
            "cd_max": float(cds.max()), "t_cd_max": float((np.argmax(cds) + 1) * deltat), "cl_max": float(cls.max()),
            "t_cl_max": float((np.argmax(cls) + 1) * deltat), "cl_min": float(cls.min()), "dp_T": float(dp),
            "outer_iterations_mean": float(np.mean(its)), "wall_s": wall}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", type=int, nargs="+", default=[4, 8])
    ap.add_argument("--deltat", type=float, nargs="+", default=[0.01])
    ap.add_argument("--T", type=float, default=8.0)
    ap.add_argument("--n-sub", type=int, default=8)
    args = ap.parse_args()
    print("# Schaefer-Turek DFG benchmark 2D-3 through libnsx (device path), reference defaults: aSIMPLE, 1e-4 abs / 1e-2 inner, implicit Euler")
    print("# published intervals: c_D max 2.93 - 2.97 | c_L max 0.47 - 0.49 | Delta p(8 s) -0.115 ... -0.105")
    print("# level  cells  DoF  deltat  steps | c_D max (at t) | c_L max (at t) | c_L min | Delta p(T) | outer its/step | wall s")
    for level in args.levels:
        for dt in args.deltat:
            r = run(level, dt, args.T, args.n_sub)
            print("%5d %6d %7d  %.4g %5d | %.4f (%.2f) | %.4f (%.2f) | %.4f | %.5f | %.1f | %.0f" % (
                r["level"], r["cells"], r["n_dofs"], r["deltat"], r["steps"], r["cd_max"], r["t_cd_max"], r["cl_max"], r["t_cl_max"], r["cl_min"],
                r["dp_T"], r["outer_iterations_mean"], r["wall_s"]), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
