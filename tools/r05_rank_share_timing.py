"""development (round 5): the kernels of ONE rank's share of a partitioned mesh, alone on the card.

    python3 tools/r05_rank_share_timing.py LEVEL WORLD RANK [steps]

The handle is built exactly as `bench.py --gpus WORLD` builds rank RANK's (deal.II's first-touch numbering of the rank's own part, internal
layout inside it, distributed mesh with ghost layers and halo plan) but with communication callbacks that do NOTHING: the all-reduce leaves
the rank's local sums, the exchange leaves the ghosts as they are.  The numbers that come out of such a solve mean nothing; the KERNELS are
the ones an 8-GPU run launches on this rank, and their HIP-event times (nsx_profile_*) are what DESIGN.md section 6's estimate is built
from: the LDS-staged SpMV's two launches (chunks without / with a ghost column), the triangular solve, the two-pass sweep, the
two-launch Schur CG.  (RCCL refuses two ranks on one device, and several processes on one card time-slice each other's kernels.)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from navierstokes_project_nm4pde_amd import nsx  # noqa: E402
from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values  # noqa: E402

level, world, rank = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2


def silent_callbacks():
    def exchange(ctx, n, ranks, send, scount, recv, rcount):
        # nobody is there to send: the ghosts get this rank's own interface values of the same vector, repeated (values of the right size
        # and sign -- the weights of the Schur product must not be zero, or its ILU(0) meets a zero pivot)
        for k in range(n):
            if rcount[k]:
                r = np.ctypeslib.as_array(recv[k], shape=(rcount[k],))
                r[:] = np.resize(np.ctypeslib.as_array(send[k], shape=(scount[k],)), rcount[k]) if scount[k] else 0.0
        return 0
    return nsx.ALLREDUCE_FN(lambda ctx, buf, count: 0), nsx.EXCHANGE_FN(exchange)


nsx.gloo_callbacks = silent_callbacks


class Args:
    ranks, numbering, ordering, schur_blocks, ranks_input, balance = 4096, "first_touch", "colour", 0, 1, "cells"


n_guess = bench.BASE_DOFS if level == bench.BASE_LEVEL else (10644763 if level == 16 else None)
ranks = bench.total_ranks(Args, n_guess) if n_guess else Args.ranks * world
mesh, dofs, tables = bench.build_problem(level, ranks, world, Args.ordering, Args.balance, Args.numbering, Args.ranks_input)
dev = nsx.Nsx(dofs, tables, bench.NU, bench.DT, device=0, rank=rank, world=world, comm="callbacks", layout=bench.layout_of(Args, dofs, world))
inlet = InletVelocity(3)
rng = np.random.default_rng(7)
dev.set_solution(1e-3 * rng.standard_normal(dofs.n_dofs))
t = 0.0
import time
wall = []
for step in range(2 * steps + 1):
    t += bench.DT
    dev.assemble(nsx.TEMAM) if step == 0 else dev.assemble_time_step(0)
    dev.apply_boundary_values(*cylinder_boundary_values(dofs, inlet, t))
    if step == steps + 1:
        dev.profile(True)   # the first `steps` solves behind the first one are timed by the wall clock WITHOUT the per-launch events, the rest with them
    t0 = time.perf_counter()
    try:
        dev.solve_time_step(nsx.YOSIDA, maxiter=4, inner_maxiter=20, check=False)
    except nsx.NsxError as e:   # no convergence is expected: the sums are this rank's alone
        print("step %d: %s" % (step, e), file=sys.stderr)
    if 1 <= step <= steps:
        wall.append(time.perf_counter() - t0)
table = dev.profile_table()
info = dev.path_info()
out = {"level": level, "n_dofs": dofs.n_dofs, "world": world, "rank": rank, "paths": info, "solve_wall_ms_unprofiled": [round(1e3 * w, 3) for w in wall],
       "spmv_F_launches_per_profiled_solve": (table.get("spmv_F", {}).get("launches", 0) / float(steps)),
       "kernels_us": {k: {"avg_us": round(1e3 * v["total_ms"] / v["launches"], 2), "launches": v["launches"],
                          "alg_GBps": round(v["bytes_per_launch"] / (1e-3 * v["total_ms"] / v["launches"]) / 1e9, 1) if v["bytes_per_launch"] > 0 else None}
                      for k, v in sorted(table.items(), key=lambda kv: -kv[1]["total_ms"]) if v["launches"] > 0 and v["total_ms"] > 0}}
print(json.dumps(out))
dev.close()
