"""development (round 5): is the difference between the fused ILU + sweep launch and the separate kernels rounding-level chaos or a bug?
Variants of the same two solves (Yosida, then aSIMPLE, tight tolerances): separate kernels, separate kernels with another grouping of
the sweep's sums (NSX_MGS_MAXWG), fused.  Prints iteration counts and solution differences."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import Problem  # noqa: E402
import navierstokes_project_nm4pde_amd.nsx as nsx  # noqa: E402
from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values  # noqa: E402


def run(p, env):
    os.environ.update(env)
    try:
        dev = p.device()
        dev.set_solution(p.smooth_velocity())
        dev.assemble(nsx.TEMAM)
        dev.apply_boundary_values(*cylinder_boundary_values(p.dofs, InletVelocity(p.dim, 2), p.deltat))
        dev.prec_initialize(0)
        src = np.random.default_rng(3).standard_normal(p.dofs.n_dofs)
        y, sv = dev.prec_vmult(0, src, inner_rtol=1e-10)
        out = [("vmult", sv, y)]
        for k in (0, 3):
            st = dev.solve_time_step(k, tol_abs=1e-10, inner_rtol=1e-8)
            out.append(("prec%d" % k, st, dev.solution_owned.copy()))
        info = dev.path_info()
        dev.close()
        return out, info
    finally:
        for k in env:
            os.environ.pop(k, None)


for dim, level, n_sub in ((3, 2, 200), (3, 2, 24)):
    p = Problem("cylinder", dim, level, n_sub=n_sub, ordering="colour")
    res = {name: run(p, env) for name, env in (("separate", {}), ("separate_maxwg40", {"NSX_MGS_MAXWG": "40"}), ("fused", {"NSX_ILU_MGS": "1"}))}
    ref = res["separate"][0]
    for name, (out, info) in res.items():
        print("case %dd level %d n_sub %d: %s (fused launches %d, sweep entries %d)" % (dim, level, n_sub, name, info["fused_launches"], info["sweep_entries_per_thread_max"]))
        for (tag, st, x), (_, st0, x0) in zip(out, ref):
            print("   %-6s outer %4d  inner F %6d (%3d solves)  inner S %6d (%3d solves)  status %d   |x - x_separate| / |x| = %.2e"
                  % (tag, st.get("outer_iterations", 0), st["inner_F_iterations"], st["n_F_solves"], st["inner_S_iterations"], st["n_S_solves"], st["status"],
                     np.abs(x - x0).max() / np.abs(x0).max()))
