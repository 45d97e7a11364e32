"""Development check: outer / inner iteration counts per time step of the bench problem for a given environment."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from navierstokes_project_nm4pde_amd import nsx  # noqa: E402
from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values  # noqa: E402

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 45
if os.environ.get("NSX_NUMBERING", "first_touch") == "first_touch":  # the bench default: deal.II's numbering handed over, layout built inside libnsx
    mesh, dofs, tables = bench.build_problem(7, 4096, 1, "colour", numbering="first_touch", ranks_input=1)
    dev = nsx.Nsx(dofs, tables, 1e-3, 2e-4, layout=(4096, nsx.COLOUR, bench.SCHUR_ROWS))
else:
    mesh, dofs, tables = bench.build_problem(7, 4096, 1, "colour", balance=os.environ.get("NSX_BALANCE", "cells"))
    dev = nsx.Nsx(dofs, tables, 1e-3, 2e-4)
    dev.set_schur_blocks(bench.schur_block_table(dofs, 0))
inlet = InletVelocity(3)
dev.set_solution(np.zeros(dofs.n_dofs))
t = 0.0
import time
outs, fs, ss = [], [], []
t_wall = 0.0
for step in range(n_steps):
    t0 = time.perf_counter()
    t += 2e-4
    dev.assemble(nsx.TEMAM) if step == 0 else dev.assemble_time_step(0)
    dev.apply_boundary_values(*cylinder_boundary_values(dofs, inlet, t))
    st = dev.solve_time_step(nsx.YOSIDA)
    if step > 0:
        t_wall += time.perf_counter() - t0
    outs.append(st["outer_iterations"]); fs.append(st["inner_F_iterations"]); ss.append(st["inner_S_iterations"])
print("outer", outs)
print("S/solve", [round(s / (o + 1), 1) for s, o in zip(ss, outs)])
print("F/solve", [round(f / (2 * o + 2), 1) for f, o in zip(fs, outs)])
print("mean outer %.2f  ms/step %.1f  ms per outer iteration %.3f" % (sum(outs) / len(outs), 1e3 * t_wall / (n_steps - 1), 1e3 * t_wall / sum(outs[1:])))
