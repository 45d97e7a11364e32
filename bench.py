#!/usr/bin/env python3
"""bench.py — time-steps/sec of the per-time-step hot path on the 3D flow-past-cylinder problem.

One "step" = NavierStokes::assemble_time_step + apply_boundary_values + solve_time_step
(reference Navier-Stokes/src/NavierStokes3D.cpp:721-724) on the ~1M-DoF P2/P1 tetrahedral mesh
(BASELINE.json configs[1]), Yosida preconditioner, reference tolerances (1e-4 abs outer, 1e-2 rel inner),
dt = 2e-4, nu = 1e-3, u_m = 9 (reference defaults).  Inputs are resident in HBM when the timed
region starts; VTU output and forces are excluded.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--level L] [--ranks R] [--schur-blocks S]

For N > 1 the driver launches one process per GPU with torch.distributed.run.  ONE mesh is partitioned over the N
GPUs (owned rows + ghost layers per rank, RCCL halo exchange of ghost DoFs inside every SpMV, RCCL all-reduce of every
dot product, per-rank ILU(0) exactly as the reference's MPI run).  Default `--scaling weak`: the mesh grows with N so
that every GPU keeps ~1M DoF (N = 8 gives the ~10M-DoF configuration of BASELINE.json configs[3]) and `value` is the
whole-job rate normalised to the N = 1 workload, value = time-steps/s x (DoF_N / DoF_1), i.e. DoF-steps/s in units of
the 1.09M-DoF problem; `--scaling strong` keeps the 1.09M-DoF mesh for every N (latency bound: one 8-byte all-reduce
per Gram-Schmidt coefficient).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

# scope name used by the library's HIP-event timer -> kernel symbol in rocprofv3 output
KERNEL_OF = {"mgs_sweep": "nsx::k_mgs", "add_and_dot": "void nsx::k_reduce<1>", "dot": "void nsx::k_reduce<0>", "spmv_F": "void nsx::k_spmv_blocked<3, 16>",
             "ilu_solve_F": "void nsx::k_ilu_solve_packed<3, 8, 8>", "ilu_solve_S": "void nsx::k_ilu_solve_packed<1, 32, 8>",
             "axpby": "nsx::k_axpby", "spmv_S": "void nsx::k_spmv_csr<32>"}


def pmc_traffic(scope):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes of this same command (FETCH_SIZE and
    WRITE_SIZE in separate passes, values in KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request, so it is doubled:
    /opt/skills/guides/MI355X_MICROARCH.md section HBM).  None when no profile is committed for the kernel."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_fetch_write_per_kernel.json")  # tools/pmc_summary.py
    try:
        with open(path) as f:
            tab = json.load(f)
        e = tab[KERNEL_OF[scope]]
        return (2.0 * e["FETCH_SIZE_KB_avg"] + e["WRITE_SIZE_KB_avg"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


def build_problem(level, ranks, world=1, ordering="colour"):
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables
    mesh = Mesh.cylinder(3, level).partition(world, max(1, ranks // world))
    return mesh, DoFs(mesh, ordering), Tables(3)


def gpu_run(dofs, tables, steps, warmup, schur_blocks, device, profile_steps=2, barrier=None, rank=0, world=1):
    import numpy as np
    from navierstokes_project_nm4pde_amd import nsx
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    nu, dt = 1e-3, 2e-4
    # NSX_BENCH_COMM=callbacks (development only): host-callback exchange over torch.distributed instead of RCCL, which lets
    # the N > 1 control flow be rehearsed with several ranks on ONE card
    dev = nsx.Nsx(dofs, tables, nu, dt, device=device, rank=rank, world=world, comm=os.environ.get("NSX_BENCH_COMM", "rccl"))
    if schur_blocks and schur_blocks < dofs.n_subdomains:
        # coarser ILU blocks for the Schur matrix: unions of consecutive virtual ranks (of this GPU)
        n_sub = dofs.n_subdomains // world
        mine = dofs.owned_p_ptr[rank * n_sub:(rank + 1) * n_sub + 1]
        stride = max(1, n_sub // max(1, schur_blocks // world))
        ptr = list(mine[::stride])
        if ptr[-1] != mine[-1]:
            ptr.append(mine[-1])
        dev.set_schur_blocks(np.array(ptr, dtype=np.int32))
    inlet = InletVelocity(3)  # test case 2, u_m = 9 (reference NavierStokes3D.hpp:37,80)
    dev.set_solution(np.zeros(dofs.n_dofs))  # u_0 = 0 (reference NavierStokes3D.hpp:200)
    t = 0.0
    stats = []

    def one_step(first):
        nonlocal t
        t += dt
        if first:
            dev.assemble(nsx.TEMAM)
        else:
            dev.assemble_time_step(0)
        bd, bv = cylinder_boundary_values(dofs, inlet, t)
        dev.apply_boundary_values(bd, bv)
        return dev.solve_time_step(nsx.YOSIDA)  # raises on non-convergence

    one_step(True)  # the first step is the full assembly (reported separately by the reference)
    for _ in range(warmup):
        one_step(False)
    if barrier:
        barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        stats.append(one_step(False))
    if barrier:
        barrier()
    elapsed = time.perf_counter() - t0
    # per-kernel HIP-event pass (separate from the throughput pass: event pairs perturb the launch stream)
    table = {}
    if profile_steps:
        dev.profile(True)
        for _ in range(profile_steps):
            one_step(False)
        table = dev.profile_table()
        dev.profile(False)
    dev.close()
    return elapsed, stats, table


def cpu_baseline(level=2, ranks=16):
    """Oracle (CPU restatement of the reference algorithm, 1 core) on a bounded sample of the same workload."""
    import numpy as np
    import oracle
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    mesh, dofs, tables = build_problem(level, ranks)
    nu, dt = 1e-3, 2e-4
    o = oracle.Oracle(dofs, tables, nu, dt)
    inlet = InletVelocity(3)
    o.assemble(oracle.TEMAM)
    bd, bv = cylinder_boundary_values(dofs, inlet, dt)
    o.apply_boundary_values(bd, bv)
    o.solve_time_step(oracle.YOSIDA)
    t0 = time.perf_counter()
    o.assemble_time_step(0)
    bd, bv = cylinder_boundary_values(dofs, inlet, 2 * dt)
    o.apply_boundary_values(bd, bv)
    st = o.solve_time_step(oracle.YOSIDA)
    el = time.perf_counter() - t0
    return dofs, tables, {"value": 1.0 / el, "unit": "time-steps/s", "cores": 1, "kind": "port",
                          "sample": "1 time step (assemble_time_step + Dirichlet + Yosida solve_time_step) of the same 3D cylinder "
                                    "problem on a %d-DoF mesh (level %d, %d ranks), oracle/nsx_oracle.c, gcc -O3, %d outer its"
                                    % (dofs.n_dofs, level, ranks, st["outer_iterations"]),
                          "sample_dofs": dofs.n_dofs, "sample_seconds": el}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10,
                    help="timed time steps (GMRES(28) needs a restart in some steps and not in others: 25 or 50 outer "
                         "iterations; ten steps average over that)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--level", type=int, default=None, help="mesh level (default: 7 ~ 1.09M DoF per GPU)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--ranks", type=int, default=4096, help="virtual MPI ranks = ILU(0) blocks of F")
    ap.add_argument("--schur-blocks", type=int, default=512, help="ILU(0) blocks of the Schur matrix")
    ap.add_argument("--ordering", choices=("colour", "first_touch"), default="colour",
                    help="velocity node order inside a virtual rank (include/nsx_host.h: nsxh_distribute_dofs_ordered)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "NSX_BENCH_DEVICE" in os.environ:  # development only: several ranks on one card
        local_rank = int(os.environ["NSX_BENCH_DEVICE"])
    import torch
    barrier = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(os.environ.get("NSX_BENCH_PG", "nccl"))  # "gloo" only for the one-card rehearsal
        def barrier():
            dist.barrier()
            torch.cuda.synchronize()
    else:
        def barrier():
            torch.cuda.synchronize()

    mode = "partitioned"
    base_level = 7
    if args.level is None:
        args.level = base_level if (world == 1 or args.scaling == "strong") else {2: 9, 4: 11, 8: 14}.get(world, int(round(7 * world ** (1 / 3.0))))
    base_ranks, base_schur = args.ranks, args.schur_blocks
    if world > 1 and args.scaling == "weak":  # same rows per virtual rank on every GPU
        args.ranks *= world
        args.schur_blocks *= world
    mesh, dofs, tables = build_problem(args.level, args.ranks, world, args.ordering)
    # profiling pass on all ranks (collective calls inside the solve must match on every rank)
    try:
        elapsed, stats, table = gpu_run(dofs, tables, args.steps, args.warmup, args.schur_blocks, local_rank,
                                        profile_steps=2, barrier=barrier, rank=rank, world=world)
        failed = ""
    except Exception as e:  # noqa: BLE001 - reported in the JSON line below
        if world == 1:
            raise
        failed = "%s: %s" % (type(e).__name__, e)
    if world > 1:
        # did every rank get through the partitioned run?  (torch's own process group, independent of libnsx's communicator)
        import torch.distributed as dist
        flag = torch.tensor([0 if failed else 1], dtype=torch.int32, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            # last resort so that the scaling run still yields a line: N independent replicas of the whole problem (weak scaling)
            mode = "replicas (partitioned run failed on a rank: %s)" % (failed or "another rank")
            # every GPU advances its own copy of the N = 1 workload
            args.level, args.ranks, args.schur_blocks = base_level, base_ranks, base_schur
            mesh, dofs, tables = build_problem(args.level, args.ranks, 1, args.ordering)
            elapsed, stats, table = gpu_run(dofs, tables, args.steps, args.warmup, args.schur_blocks, local_rank,
                                            profile_steps=2 if rank == 0 else 0, barrier=barrier, rank=0, world=1)
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    # partitioned: the whole job advances `steps` steps of ONE problem; replicas: every rank advances its own copy
    raw_steps_per_s = args.steps / elapsed
    base_dofs = 1089643  # level-7 mesh, the N = 1 workload
    if mode != "partitioned":
        steps_per_s = world * raw_steps_per_s
    elif world > 1 and args.scaling == "weak":
        steps_per_s = raw_steps_per_s * dofs.n_dofs / base_dofs
    else:
        steps_per_s = raw_steps_per_s
    outer = sum(s["outer_iterations"] for s in stats)
    t_solve = sum(s["t_solve"] for s in stats)
    # roofline of the dominant kernel (by summed HIP-event time over the profiled steps)
    kernels = {}
    for k, v in table.items():
        if v["launches"] == 0:
            continue
        avg_s = v["total_ms"] * 1e-3 / v["launches"]
        kernels[k] = {"launches_per_step": v["launches"] / 2.0, "avg_us": avg_s * 1e6,
                      "alg_GBps": (v["bytes_per_launch"] / avg_s / 1e9) if avg_s > 0 and v["bytes_per_launch"] > 0 else None,
                      "share": v["total_ms"]}
    tot = sum(v["share"] for v in kernels.values()) or 1.0
    for v in kernels.values():
        v["share"] = v["share"] / tot
    dom = max((k for k in kernels if kernels[k]["alg_GBps"]), key=lambda k: kernels[k]["share"]) if kernels else None
    roof = None
    if dom:
        a = kernels[dom]["alg_GBps"]
        roof = {"kernel": dom, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                "traffic": pmc_traffic(dom), "algorithmic_bytes": table[dom]["bytes_per_launch"],
                "avg_us": kernels[dom]["avg_us"], "share_of_kernel_time": kernels[dom]["share"]}
        if "spmv_F" in kernels and kernels["spmv_F"]["alg_GBps"]:
            roof["spmv_F_GBps"] = kernels["spmv_F"]["alg_GBps"]
            roof["spmv_F_frac"] = kernels["spmv_F"]["alg_GBps"] / HBM_PEAK_GBS
    out = {
        "metric": "time-steps/sec (assemble_time_step + solve_time_step), 3D flow past a cylinder, P2/P1, Yosida",
        "value": steps_per_s, "unit": "time-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling if mode == "partitioned" else "weak", "vs_baseline": None,
        "time_steps_per_s_of_this_mesh": raw_steps_per_s,
        "dtype": "f64", "data": "synthetic (block-structured tetrahedral cylinder mesh, u0 = 0, reference inlet profile)",
        "config": {"workload": "3D flow-past-cylinder, P2/P1 (reference FE_SimplexP), %d DoF, %d cells, dt=2e-4, nu=1e-3, u_m=9, "
                               "GMRES(1e-4 abs)+Yosida(inner 1e-2), ILU(0) per rank with %d ranks (Schur: %d blocks), %s node order inside a rank"
                               % (dofs.n_dofs, dofs.n_cells, args.ranks, args.schur_blocks, args.ordering),
                   "n_dofs": dofs.n_dofs, "n_cells": dofs.n_cells,
                   "parallelism": "mesh partitioned over %d GPU(s): RCCL ghost exchange + dot-product all-reduce" % world
                   if mode == "partitioned" else mode},
        "gmres_outer_iters_per_step": outer / max(1, len(stats)),
        "gmres_outer_iters_per_sec": outer / t_solve if t_solve > 0 else None,
        "inner_F_iters_per_step": sum(s["inner_F_iterations"] for s in stats) / max(1, len(stats)),
        "inner_S_iters_per_step": sum(s["inner_S_iterations"] for s in stats) / max(1, len(stats)),
        "roofline": roof,
        "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in kernels.items()},
    }
    if world == 1 and not args.no_cpu:
        import multiprocessing
        sd, st, cb = cpu_baseline()
        # the same sample on the GPU, for a like-for-like ratio
        e2, _, _ = gpu_run(sd, st, 3, 1, 0, local_rank, profile_steps=0)
        cb["gpu_same_sample_steps_per_s"] = 3 / e2
        cb["host_cores_available"] = multiprocessing.cpu_count()
        cb["estimate_full_workload_steps_per_s"] = cb["value"] * cb["sample_dofs"] / dofs.n_dofs
        out["cpu_baseline"] = cb
    print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
