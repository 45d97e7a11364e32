#!/usr/bin/env python3
"""bench.py — time-steps/sec of the per-time-step hot path on the 3D flow-past-cylinder problem.

One "step" = NavierStokes::assemble_time_step + apply_boundary_values + solve_time_step
(reference Navier-Stokes/src/NavierStokes3D.cpp:721-724) on the ~1M-DoF P2/P1 tetrahedral mesh
(BASELINE.json configs[1]), Yosida preconditioner, reference tolerances (1e-4 abs outer, 1e-2 rel inner),
dt = 2e-4, nu = 1e-3, u_m = 9 (reference defaults).  Inputs are resident in HBM when the timed
region starts; VTU output and forces are excluded.  The run starts from u0 = 0 with the inlet switched on
impulsively, as the reference does; the first step (full assembly) and --spinup further steps (default 20, SURVEY 8d:
"timing uses steps after a fixed warm-up") prepare the state, then come W warm-up steps and exactly K timed steps (default
K = 100 on one GPU; under a launcher K = 30 behind 10 preparation steps).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--level L] [--ranks R] [--numbering first_touch|preordered] [--ranks-input Q]

What the caller hands over (default): deal.II's OWN numbering -- first-touch order of distribute_dofs + component_wise on Q ranks
per GPU (Q = 1: the serial reference, reference NavierStokes3D.cpp:16-19,58-69) -- and libnsx lays the nodes out behind the C-ABI
(nsx_set_internal_layout: R virtual ranks, colour order, Schur blocks of <= 96 rows).  --numbering preordered hands over the
front-end's R-rank colour numbering instead (rounds 1-3); the two are bitwise the same computation (tests/test_gpu_layout.py).

N > 1: one process per GPU under torch.distributed.run (the driver's launch line); `python bench.py --gpus N` without
that launcher starts it as a child process.  ONE mesh is partitioned over the N GPUs (owned rows + ghost layers per
rank, RCCL halo exchange of ghost DoFs inside every SpMV, RCCL all-reduce of every dot product, per-rank ILU(0) exactly
as the reference's MPI run).  `value` is ALWAYS plain time-steps/s of the 1 089 643-DoF mesh of the N = 1 line ("scaling":
"strong": value(N) / value(1) is the speed-up of that mesh).  The same invocation also partitions the 10 644 763-DoF mesh
(level 16, BASELINE.json configs[3]) over the N GPUs and reports it under "strong_10M" beside its committed one-GPU base
(profiles/r04_strong_10M_one_gpu.json): the >= 6x of SURVEY 8(d) is read from that entry.  A rank that fails in the headline run
makes the whole job exit non-zero; the strong_10M leg runs after the headline is measured, under a deadline, and a failure there is
reported inside "strong_10M" instead of discarding the measured line.

Other modes (not the driver's): --comm rccl1 (one GPU, 1-rank RCCL communicator: the distributed code path -- one launch + one
ncclAllReduce per reduction -- timed on one card), --cpu-only (the cpu_baseline leg alone), --layout-table FILE (iteration counts
and step times of the layouts a caller can hand over; its committed result is embedded in every bench line as
"preconditioner_layouts").  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BASE_LEVEL, BASE_DOFS = 7, 1089643  # the N = 1 workload: level-7 cylinder mesh
BIG_LEVEL, BIG_DOFS = 16, 10644763  # the "~10 M DoF" mesh of BASELINE.json configs[3] (level 14 has 7.3 M)
if os.environ.get("NSX_BENCH_BIG_LEVEL"):  # development only: rehearse the strong_10M leg's control flow on a small mesh
    BIG_LEVEL, BIG_DOFS = int(os.environ["NSX_BENCH_BIG_LEVEL"]), None
if os.environ.get("NSX_BENCH_BIG_SCHEDULE"):  # development only (one-card rehearsals): "steps,warmup,spinup" of the strong_10M leg; the line says what ran
    _BIG_SCHEDULE = tuple(int(v) for v in os.environ["NSX_BENCH_BIG_SCHEDULE"].split(","))
else:
    _BIG_SCHEDULE = None
BIG_STEPS, BIG_WARMUP, BIG_SPINUP = _BIG_SCHEDULE or (6, 1, 3)  # schedule of the strong_10M leg: the same on one GPU (committed base) and on N (short: at N = 2 a step of
                                             # this mesh still takes seconds, and the whole invocation has to stay within minutes)
PMC_PROFILE = "profiles/r05_pmc_fetch_write_per_kernel.json"
LAYOUT_PROFILE = "profiles/r05_layout_iterations.json"
BIG_BASE_PROFILE = "profiles/r05_strong_10M_one_gpu.json"
STEP_HISTORY = "profiles/r05_step_history.txt"
NU, DT = 1e-3, 2e-4

# scope name used by the library's HIP-event timer -> kernel symbol in rocprofv3 output
KERNEL_OF = {"mgs_sweep": "void nsx::k_mgs_one<8, 10, false>", "add_and_dot": "void nsx::k_reduce<1>", "dot": "void nsx::k_reduce<0>", "spmv_F": "void nsx::k_spmv_blocked<3, 16>",
             "ilu_solve_F": "void nsx::k_ilu_solve_lanes<3, 2, 8>", "ilu_solve_S": "nsx::k_ilu_apply_dense",
             "axpby": "nsx::k_axpby", "spmv_S": "void nsx::k_spmv_csr<32>", "cg_S": "void nsx::k_cg_schur<6, true>"}


def pmc_live(argv_tail, log=sys.stderr):
    """HBM counters of THIS invocation's workload: two child runs of this file under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE
    in passes of their own, kernel trace only: /opt/skills/guides/MI355X_MICROARCH.md section HBM), one timed step each, started
    BEFORE this process touches the GPU.  Returns (table, outcome): table = {kernel symbol: {"launches", "FETCH_SIZE_KB_avg",
    "WRITE_SIZE_KB_avg"}} or None; outcome = {"status": "ok" | "skipped" | "failed" | "timeout" | "error", per-pass rc / seconds /
    stderr tail}.  A pass that dies on a signal or overruns is REPORTED in the bench line (a crash of the library under the
    profiler must not pass as "quote the committed profile")."""
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, {"status": "skipped", "why": "rocprofv3 not on PATH"}
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, {"status": "skipped", "why": "already running under a profiler"}
    out, tmp, outcome = {}, None, {"status": "ok", "passes": []}
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_summary
        tmp = tempfile.mkdtemp(prefix="nsx_pmc_", dir="/tmp")
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            t0 = time.time()
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "1", "--warmup", "1", "--spinup", "2", "--no-cpu", "--profile-steps", "0", "--pmc", "off"] + argv_tail
            # a process group of its own: a pass that overruns is ended as a whole (the profiler AND the program under it)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
            rec = {"counter": counter}
            outcome["passes"].append(rec)
            try:
                _, err = proc.communicate(timeout=150)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)
                proc.communicate()
                rec.update(rc=None, seconds=time.time() - t0, note="took more than 150 s: process group ended")
                outcome["status"] = "timeout"
                print("bench.py: rocprofv3 --pmc %s pass took more than 150 s: ended" % counter, file=log)
                return None, outcome
            rec.update(rc=proc.returncode, seconds=round(time.time() - t0, 1))
            if proc.returncode != 0:
                rec["stderr_tail"] = err.decode(errors="replace")[-400:]
                outcome["status"] = "failed" if proc.returncode > 0 else "signal"
                print("bench.py: rocprofv3 --pmc %s pass failed (rc %d): %s" % (counter, proc.returncode, rec["stderr_tail"]), file=log)
                return None, outcome
            for k, counters in pmc_summary.collect(d).items():
                for c, (total, ids) in counters.items():
                    e = out.setdefault(k, {})
                    e["launches"] = len(ids)
                    e[c + "_KB_avg"] = total / max(1, len(ids))
            print("bench.py: rocprofv3 --pmc %s pass: %.0f s" % (counter, time.time() - t0), file=log, flush=True)
    except Exception as e:  # noqa: BLE001 -- the counters are an extra: the bench line then quotes the committed profile AND says why
        print("bench.py: live PMC passes not available: %s: %s" % (type(e).__name__, e), file=log)
        outcome.update(status="error", error="%s: %s" % (type(e).__name__, e))
        return None, outcome
    finally:
        if tmp:
            shutil.rmtree(tmp, ignore_errors=True)
    return out, outcome


def pmc_traffic(scope, live=None):
    """HBM-side bytes per launch of the kernel behind `scope` (FETCH_SIZE and WRITE_SIZE in separate passes, values in KiB; gfx950
    FETCH_SIZE counts 64 B per 128-B request, so it is doubled: /opt/skills/guides/MI355X_MICROARCH.md section HBM).  From the
    live passes of this invocation (pmc_live) when there are any, else the constant of the COMMITTED passes of this same command;
    (None, None) when neither has the kernel."""
    def of(tab):
        e = tab[KERNEL_OF[scope]]
        return (2.0 * e["FETCH_SIZE_KB_avg"] + e["WRITE_SIZE_KB_avg"]) * 1024.0
    if live:
        try:
            return of(live), "live"
        except KeyError:
            pass
    for rel in (PMC_PROFILE, "profiles/r04_pmc_fetch_write_per_kernel.json"):
        try:
            with open(os.path.join(ROOT, rel)) as f:
                return of(json.load(f)), rel
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def build_problem(level, ranks, world=1, ordering="colour", balance="cells", numbering="preordered", ranks_input=1):
    """numbering = "first_touch": what a deal.II caller hands over -- first-touch order on `ranks_input` ranks per GPU (the
    virtual ranks are then built inside libnsx: see layout_of); "preordered": the front-end's `ranks`-rank numbering with the
    node order `ordering` inside a rank."""
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables
    if numbering == "first_touch":
        mesh = Mesh.cylinder(3, level).partition(world, max(1, ranks_input), balance=balance)
        return mesh, DoFs(mesh, "first_touch"), Tables(3)
    mesh = Mesh.cylinder(3, level).partition(world, max(1, ranks // world), balance=balance)
    return mesh, DoFs(mesh, ordering), Tables(3)


def total_ranks(args, n_dofs):
    """virtual ranks over the whole mesh: --ranks is quoted for the 1.09 M-DoF mesh, larger meshes keep its ~85 rows per ILU block"""
    return max(1, int(round(args.ranks * n_dofs / float(BASE_DOFS))))


def layout_of(args, dofs, world=1):
    """(virtual ranks per GPU, node order, Schur block rows) for nsx_set_internal_layout, or None with --numbering preordered"""
    if args.numbering != "first_touch":
        return None
    from navierstokes_project_nm4pde_amd import nsx
    return (max(1, total_ranks(args, dofs.n_dofs) // world), {"colour": nsx.COLOUR, "first_touch": nsx.FIRST_TOUCH, "colour_all": nsx.COLOUR_ALL}[args.ordering],
            SCHUR_ROWS if not args.schur_blocks else 0)


def transfer_state(src_dofs, x, dst_dofs):
    """The same finite-element function in another DoF numbering of the same mesh (other rank count / node order):
    nodes are matched through their support points, which are bitwise equal in both numberings."""
    import numpy as np
    dim = src_dofs.dim
    out = np.empty_like(x)
    for lo, hi, stride in ((0, src_dofs.n_u, dim), (src_dofs.n_u, src_dofs.n_dofs, 1)):
        ps, pd = src_dofs.support_points[lo:hi:stride], dst_dofs.support_points[lo:hi:stride]
        ks, kd = np.lexsort(ps.T[::-1]), np.lexsort(pd.T[::-1])
        assert (ps[ks] == pd[kd]).all(), "the two DoF tables do not describe the same mesh"
        perm = np.empty(len(ks), dtype=np.int64)  # node of src that sits at dst node i
        perm[kd] = ks
        for c in range(stride):
            out[lo + c:hi:stride] = x[lo + c:hi:stride][perm]
    return out


SCHUR_ROWS = 96  # default size limit of the Schur ILU blocks (rows): blocks of at most 96 rows keep their explicit inverses in registers


def schur_block_table(dofs, schur_blocks, rank=0, world=1, max_rows=SCHUR_ROWS):
    """coarser ILU blocks for the Schur matrix: unions of consecutive virtual ranks (of this GPU).  schur_blocks = 0: as many
    consecutive ranks per block as fit `max_rows` pressure rows (blocks of 86 - 96 rows at 4096 ranks: ~505 blocks of equal
    size instead of 512 blocks of 55 - 173 rows); schur_blocks > 0: that many blocks of equal rank count (round 1 / 2 layout)."""
    import numpy as np
    from navierstokes_project_nm4pde_amd.frontend import merge_ranks
    n_sub = dofs.n_subdomains // world
    mine = dofs.owned_p_ptr[rank * n_sub:(rank + 1) * n_sub + 1]
    if not schur_blocks:
        return merge_ranks(mine, max_rows)
    stride = max(1, n_sub // max(1, schur_blocks // world))
    ptr = list(mine[::stride])
    if ptr[-1] != mine[-1]:
        ptr.append(mine[-1])
    return np.array(ptr, dtype=np.int32)


_last_beat = [time.time()]


def beat(msg, every=30.0):
    """a line on stderr at most every `every` seconds: a long leg (the 10 M-DoF mesh spends minutes in set-up and ~8 s per step) must
    not look hung to whoever watches the job"""
    now = time.time()
    if now - _last_beat[0] >= every:
        _last_beat[0] = now
        print("bench.py: " + msg, file=sys.stderr, flush=True)


def gpu_run(dofs, tables, steps, warmup, schur_blocks, device, profile_steps=5, barrier=None, rank=0, world=1, want_state=False, spinup=0,
            layout=None, comm_single=False, hoisted_twin=False):
    """first step + `spinup` steps (state preparation) + warmup + `steps` timed steps with the REFERENCE's schedule of the
    preconditioner set-up (Schur product, ILU(S) and block inverses rebuilt in every step, Preconditioners.hpp:358-362: NSX_SCHUR_CACHE=0
    unless the environment says otherwise; a handle with a communicator rebuilds anyway) (+ the same timed steps once more with those
    products kept while their inputs are bit-identical, + a separate per-kernel HIP-event pass).  The handle is closed on every path: a
    failure must not leave a communicator or a second copy of the problem behind."""
    import numpy as np
    from navierstokes_project_nm4pde_amd import nsx
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    # NSX_BENCH_COMM=callbacks (development only): host-callback exchange over torch.distributed instead of RCCL, which lets
    # the N > 1 control flow be rehearsed with several ranks on ONE card
    t_setup = time.perf_counter()
    cache_env = os.environ.get("NSX_SCHUR_CACHE")
    if cache_env is None:
        os.environ["NSX_SCHUR_CACHE"] = "0"   # read by the library at every preconditioner initialisation
    beat("%d DoF: handing the mesh over (set-up products, internal layout)" % dofs.n_dofs, every=0.0 if dofs.n_dofs > 3e6 else 30.0)
    dev = nsx.Nsx(dofs, tables, NU, DT, device=device, rank=rank, world=world, comm=os.environ.get("NSX_BENCH_COMM", "rccl"), layout=layout)
    try:
        if not layout and schur_blocks < dofs.n_subdomains:
            dev.set_schur_blocks(schur_block_table(dofs, schur_blocks, rank, world))
        if comm_single:
            dev.comm_init_single()  # 1-rank RCCL communicator: every reduction becomes a launch + ncclAllReduce, the sweep keeps its collective inside, the Schur CG takes two launches per iteration
        inlet = InletVelocity(3)  # test case 2, u_m = 9 (reference NavierStokes3D.hpp:37,80)
        dev.set_solution(np.zeros(dofs.n_dofs))  # u_0 = 0 (reference NavierStokes3D.hpp:200)
        t = 0.0
        stats = []

        def one_step(first):
            nonlocal t
            t += DT
            if first:
                dev.assemble(nsx.TEMAM)
            else:
                dev.assemble_time_step(0)
            bd, bv = cylinder_boundary_values(dofs, inlet, t)
            dev.apply_boundary_values(bd, bv)
            st = dev.solve_time_step(nsx.YOSIDA)  # raises on non-convergence
            beat("t = %.4f: %d outer iterations in the last step" % (t, st["outer_iterations"]))
            return st

        one_step(True)  # the first step is the full assembly (reported separately by the reference)
        t_setup = time.perf_counter() - t_setup
        for _ in range(spinup + warmup):
            one_step(False)
        twin_state = (dev.solution_owned, t) if (hoisted_twin and world == 1 and not comm_single and cache_env is None) else None
        if barrier:
            barrier()
        c0 = dev.comm_counters() if (world > 1 or comm_single) else (0, 0)
        t0 = time.perf_counter()
        for _ in range(steps):
            stats.append(one_step(False))
        if barrier:
            barrier()
        elapsed = time.perf_counter() - t0
        if stats:
            stats[0]["setup_and_first_step_s"] = t_setup
            stats[0]["layout"] = dev.layout_info()
            stats[0]["paths"] = dev.path_info()
            if world > 1:  # every rank says which paths it took / would take under RCCL (a rehearsal over host callbacks runs the two-pass sweep)
                print("bench.py: rank %d of %d, %d DoF: paths %s" % (rank, world, dofs.n_dofs, json.dumps(stats[0]["paths"])), file=sys.stderr, flush=True)
            stats[0]["schur_cache"] = os.environ.get("NSX_SCHUR_CACHE") != "0" and world == 1 and not comm_single
        if (world > 1 or comm_single) and stats:  # collectives this rank issued per timed step (nsx_comm_counters)
            c1 = dev.comm_counters()
            stats[0]["allreduces_per_step"] = (c1[0] - c0[0]) / float(steps)
            stats[0]["ghost_exchanges_per_step"] = (c1[1] - c0[1]) / float(steps)
        state = (dev.gather_solution() if world > 1 else dev.solution_owned, t) if want_state else None
        if twin_state is not None and stats:
            # the hoisted twin on the SAME timed steps: the state in front of the timed region is restored and the steps are run again
            # with the Schur product, its ILU(0) factors and the block inverses kept while their inputs are bit for bit unchanged
            # (Yosida: always) -- all sums are fixed-order, so they walk through the same iteration counts
            t_end = t
            dev.set_solution(twin_state[0])
            t = twin_state[1]
            os.environ["NSX_SCHUR_CACHE"] = "1"
            try:
                torch_sync = barrier or (lambda: None)
                torch_sync()
                t0 = time.perf_counter()
                kept = [one_step(False) for _ in range(steps)]
                torch_sync()
                el_kept = time.perf_counter() - t0
            finally:
                os.environ["NSX_SCHUR_CACHE"] = "0"
            stats[0]["hoisted"] = {"elapsed": el_kept, "t_prec": sum(s["t_prec"] for s in kept) / len(kept),
                                   "outer": [s["outer_iterations"] for s in kept]}
            assert abs(t - t_end) < 1e-12
        # per-kernel HIP-event pass (separate from the throughput pass: event pairs perturb the launch stream)
        table, prof_stats = {}, []
        if profile_steps:
            dev.profile(True)
            for _ in range(profile_steps):
                prof_stats.append(one_step(False))
            table = dev.profile_table()
            dev.profile(False)
        if stats:
            stats[0]["persistent_state"] = dev.persistent_state()
        return elapsed, stats, table, prof_stats, state
    finally:
        dev.close()
        if cache_env is None:
            os.environ.pop("NSX_SCHUR_CACHE", None)


def cpu_step_from_state(dofs, tables, x, t_state, threads, compact=False):
    """One time step (assemble_time_step + Dirichlet values + Yosida solve_time_step) of the oracle from the velocity /
    pressure state `x` (numbering of `dofs`) at time t_state.  Untimed before it: orc_create and the first full assembly
    (mass, stiffness, B blocks: NavierStokes::assemble), which the reference also runs once per run."""
    import oracle
    from navierstokes_project_nm4pde_amd.problem import InletVelocity, cylinder_boundary_values
    t0 = time.perf_counter()
    o = oracle.Oracle(dofs, tables, NU, DT, threads=threads, compact=compact)
    t_create = time.perf_counter() - t0
    inlet = InletVelocity(3)
    t0 = time.perf_counter()
    o.assemble(oracle.TEMAM)  # from u = 0: convection matrix = 0, exactly what assemble_time_step expects to subtract
    t_first = time.perf_counter() - t0
    o.solution[:] = x
    o.solution_owned[:] = x
    t0 = time.perf_counter()
    o.assemble_time_step(0)
    bd, bv = cylinder_boundary_values(dofs, inlet, t_state + DT)
    o.apply_boundary_values(bd, bv)
    t_asm = time.perf_counter() - t0
    st = o.solve_time_step(oracle.YOSIDA)
    el = time.perf_counter() - t0
    if st["status"]:
        raise RuntimeError("cpu_baseline: the oracle's solve did not converge (status %d)" % st["status"])
    return {"seconds": el, "threads": o.threads, "outer": st["outer_iterations"], "inner_F": st["inner_F_iterations"],
            "inner_S": st["inner_S_iterations"], "t_assemble": t_asm, "t_prec": st["t_prec"], "t_solve": st["t_solve"],
            "untimed_create_s": t_create, "untimed_first_assembly_s": t_first}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


NATIVE_FLAGS = ["-O3", "-march=native", "-std=c99", "-fPIC", "-fopenmp", "-Wno-unknown-pragmas"]
PORTABLE_FLAGS = "-O3 -march=x86-64-v3 -std=c99 -fPIC -ffp-contract=off -fopenmp (in-tree oracle/liboracle_mt.so, the parity tests' build)"


def native_oracle(log=sys.stderr):
    """the CPU baseline is compiled `-O3 -march=native` ON the machine it is timed on.  The in-tree
    liboracle_mt.so is the portable build the parity tests use (x86-64-v3, no FP contraction); here the same source is compiled
    for this host's cores into a scratch directory.  Returns the flags string that describes the library actually used."""
    import shutil
    import tempfile
    import oracle
    gcc = shutil.which("gcc")
    if not gcc:
        return PORTABLE_FLAGS + " [no gcc on this host for a native build]"
    d = tempfile.mkdtemp(prefix="nsx_oracle_native_", dir="/tmp")
    so = os.path.join(d, "liboracle_mt_native.so")
    cmd = [gcc] + NATIVE_FLAGS + ["-shared", "-o", so, os.path.join(ROOT, "oracle", "nsx_oracle.c"), "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print("bench.py: native build of the oracle failed (%s): %s" % (" ".join(cmd), r.stderr[-300:]), file=log)
        return PORTABLE_FLAGS + " [native build failed]"
    oracle.use_mt_library(so)
    return "gcc " + " ".join(NATIVE_FLAGS) + " (oracle/nsx_oracle.c compiled on this host for the baseline leg)"


def cpu_baseline(gpu_dofs, gpu_state, device, small_level=5):
    """The stated CPU baseline, timed on this box's host cores in this run:
    (i) ALL usable cores on the bench workload itself (same mesh, same state, R = cores ranks as `mpirun -n cores` of
        the reference would have, first-touch numbering): the oracle's source built with OpenMP, -O3 -march=native;
    (i') "best CPU": the same with compact storage of block (0,0);
    (ii) ONE core, the serial reference-shaped restatement (oracle/liboracle.so, R = 1) on the largest mesh that
        finishes in about a minute, with the GPU on that same small problem beside it."""
    import oracle
    flags = native_oracle()
    cores = oracle.usable_cores()
    model = cpu_model()
    x, t_state = gpu_state
    # (i) all cores, full workload
    mesh, dofs, tables = build_problem(BASE_LEVEL, cores, 1, "first_touch")
    r = cpu_step_from_state(dofs, tables, transfer_state(gpu_dofs, x, dofs), t_state, cores)
    out = {"value": 1.0 / r["seconds"], "unit": "time-steps/s", "cores": r["threads"], "cpu_model": model, "compiler_flags": flags, "kind": "port",
           "sample": "1 time step (assemble_time_step + Dirichlet + Yosida solve_time_step) of the bench workload itself: the %d-DoF mesh, "
                     "started from the GPU run's state at t = %.4f, %d MPI-rank ILU blocks (first-touch numbering, as `mpirun -n %d`), "
                     "oracle/nsx_oracle.c with OpenMP on %d threads of %s; %d outer / %d inner-F / %d inner-S iterations"
                     % (dofs.n_dofs, t_state, cores, cores, r["threads"], model, r["outer"], r["inner_F"], r["inner_S"]),
           "sample_dofs": dofs.n_dofs, "sample_seconds": r["seconds"], "phases_s": {k: r[k] for k in ("t_assemble", "t_prec", "t_solve")},
           "untimed_setup_s": {"orc_create": r["untimed_create_s"], "first_assembly": r["untimed_first_assembly_s"]}}
    # (i') "best CPU" (BASELINE.md section 2 (ii)): the same step, same ranks, same threads, with the products with system(0,0) and
    # its per-rank ILU(0) on the scalar P2 operator (compact storage) instead of the reference's padded dim x dim couplings --
    # NOT the reference's layout, but what a CPU code that exploited the same structure as the device library would run
    rb = cpu_step_from_state(dofs, tables, transfer_state(gpu_dofs, x, dofs), t_state, cores, compact=True)
    out["best_cpu"] = {"value": 1.0 / rb["seconds"], "unit": "time-steps/s", "cores": rb["threads"], "cpu_model": model, "compiler_flags": flags, "kind": "port",
                       "sample": "the same step with compact storage of block (0,0) (oracle.Oracle(compact=True): scalar P2 operator for the F products and "
                                 "ILU(0) solves; assembly, Schur product and rectangular blocks as in the reference): %d outer / %d inner-F / %d inner-S iterations"
                                 % (rb["outer"], rb["inner_F"], rb["inner_S"]),
                       "sample_seconds": rb["seconds"], "phases_s": {k: rb[k] for k in ("t_assemble", "t_prec", "t_solve")}}
    del mesh, dofs
    # (ii) one core, reference-shaped, bounded sample.  A short GPU run on that small mesh provides the state to start from
    # (512 virtual ranks there: the state depends on the layout only through the solver tolerance)
    _, gd, gt = build_problem(small_level, 512, 1, "colour")
    e2, _, _, _, s_state = gpu_run(gd, gt, 3, 1, 0, device, profile_steps=0, want_state=True)
    _, sd, st = build_problem(small_level, 1, 1, "first_touch")
    r1 = cpu_step_from_state(sd, st, transfer_state(gd, s_state[0], sd), s_state[1], 1)
    out["one_core"] = {"value": 1.0 / r1["seconds"], "unit": "time-steps/s", "cores": 1, "cpu_model": model,
                       "compiler_flags": "gcc -O3 -march=x86-64-v3 -std=c99 -ffp-contract=off (in-tree oracle/liboracle.so)", "kind": "port",
                       "sample": "the same step on the %d-DoF mesh (level %d), 1 rank (the serial reference's layout), oracle/nsx_oracle.c, "
                                 "%d outer / %d inner-F iterations" % (sd.n_dofs, small_level, r1["outer"], r1["inner_F"]),
                       "sample_dofs": sd.n_dofs, "sample_seconds": r1["seconds"],
                       "gpu_same_mesh_512_virtual_ranks_steps_per_s": 3 / e2}
    return out


def layout_table(device, out_path, steps=4):
    """What a caller can hand over, at full size: (caller's ranks, caller's node order, libnsx's internal layout) -> iteration counts
    of the Yosida-preconditioned solve and the step time.  The first two rows are the literal drop-in cases (deal.II's own
    first-touch numbering on 1 rank / on 8 ranks) with nsx_set_internal_layout; then the front-end's pre-ordered 4096-rank numbering
    (rounds 1-3, bitwise the same computation as row 1); then what the caller's own layouts cost WITHOUT the internal layout
    (level-per-launch ILU: one launch per dependency level)."""
    from navierstokes_project_nm4pde_amd import nsx
    rows = []
    only = os.environ.get("NSX_LAYOUT_ONLY")  # e.g. "1:first_touch:4096": measure these rows only
    cases = [(1, "first_touch", 4096), (8, "first_touch", 4096), (4096, "colour", 0), (8, "colour_all", 0), (1, "colour_all", 0), (8, "first_touch", 0)]
    for ranks, ordering, virtual in cases:
        if only and "%d:%s:%d" % (ranks, ordering, virtual) not in only.split(","):
            continue
        mesh, dofs, tables = build_problem(BASE_LEVEL, ranks, 1, ordering)
        t0 = time.perf_counter()
        n = steps if (virtual or ranks > 8) else 2
        layout = (virtual, nsx.COLOUR, SCHUR_ROWS) if virtual else None
        elapsed, stats, _, _, _ = gpu_run(dofs, tables, n, 1, 0 if ranks > 8 else ranks, device, profile_steps=0, layout=layout)
        outer = sum(s["outer_iterations"] for s in stats)
        row = {"ranks_of_the_caller": ranks, "node_order_of_the_caller": ordering, "internal_layout": ("%d virtual ranks, colour order, Schur blocks <= %d rows"
               % (virtual, SCHUR_ROWS)) if virtual else None, "n_dofs": dofs.n_dofs, "steps": n,
               "outer_per_step": outer / n, "inner_F_per_step": sum(s["inner_F_iterations"] for s in stats) / n,
               "inner_S_per_step": sum(s["inner_S_iterations"] for s in stats) / n,
               "ms_per_step": 1e3 * elapsed / n, "ms_per_outer_iteration": 1e3 * elapsed / max(1, outer),
               "setup_and_first_step_s": stats[0]["setup_and_first_step_s"], "wall_s": time.perf_counter() - t0}
        rows.append(row)
        print("[layout] %s" % json.dumps(row), file=sys.stderr, flush=True)
        del mesh, dofs
    doc = {"what": "per time step, 3D cylinder level 7 (1 089 643 DoF), Yosida, reference tolerances, steps 3.. of the run (after the first step + 1 warm-up step)",
           "command": "python3 bench.py --layout-table %s" % out_path, "rows": rows}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    return doc


def committed_layouts():
    try:
        with open(os.path.join(ROOT, LAYOUT_PROFILE)) as f:
            doc = json.load(f)
        return {"source": LAYOUT_PROFILE + " (measured on MI355X by `bench.py --layout-table`, not in this run)", "rows": doc["rows"]}
    except (OSError, KeyError, ValueError):
        return None


def long_run_mean_outer():
    """mean outer GMRES iterations per time step over the committed long run (tools/step_history.py, several hundred consecutive
    steps from u0 = 0): the timed window of a bench run is a sample of that chaotic sequence"""
    import re
    try:
        with open(os.path.join(ROOT, STEP_HISTORY)) as f:
            m = re.search(r"^mean outer ([0-9.]+)", f.read(), re.M)
        return float(m.group(1)) if m else None
    except OSError:
        return None


def committed_big_base():
    try:
        with open(os.path.join(ROOT, BIG_BASE_PROFILE)) as f:
            d = json.load(f)
        return {k: d.get(k) for k in ("value", "ms_per_step", "gmres_outer_iters_per_step", "ms_per_outer_iteration", "steps", "warmup", "spinup_steps")} | \
            {"n_dofs": d["config"]["n_dofs"], "source": BIG_BASE_PROFILE + " (one MI355X, `python3 bench.py --gpus 1 --level %d --steps %d --warmup %d --spinup %d --no-cpu`)"
             % (BIG_LEVEL, BIG_STEPS, BIG_WARMUP, BIG_SPINUP)}
    except (OSError, KeyError, ValueError):
        return None


def summarise_kernels(table, prof_stats):
    n_prof = max(1, len(prof_stats))
    kernels = {}
    for k, v in table.items():
        if v["launches"] == 0:
            continue
        avg_s = v["total_ms"] * 1e-3 / v["launches"]
        kernels[k] = {"launches_per_step": v["launches"] / float(n_prof), "avg_us": avg_s * 1e6,
                      "alg_GBps": (v["bytes_per_launch"] / avg_s / 1e9) if avg_s > 0 and v["bytes_per_launch"] > 0 else None,
                      "share": v["total_ms"]}
    tot = sum(v["share"] for v in kernels.values()) or 1.0
    for v in kernels.values():
        v["share"] = v["share"] / tot
    return kernels, tot / n_prof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed time steps.  GMRES(28) needs a restart in some steps and not in others (17 - 28 or 40 - 50 outer "
                         "iterations) and restart steps come in runs, so windows of 20 steps average anything from 20 to 36 outer "
                         "iterations; 100 steps are within ~10 %% of the 200- and 300-step means (profiles/r0*_step_history.txt); "
                         "see gmres_outer_iters_per_step / ms_per_outer_iteration / value_long_run.  Default: 100 on one GPU, 30 under a launcher")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spinup", type=int, default=None,
                    help="untimed time steps that prepare the state before the warm-up: the reference starts from u0 = 0 with the inlet "
                         "switched on impulsively, and SURVEY 8d times steps after a fixed warm-up of ~20 steps past that transient")
    ap.add_argument("--level", type=int, default=None, help="mesh level (default: 7 = 1 089 643 DoF; 16 = 10 644 763 DoF)")
    ap.add_argument("--ranks", type=int, default=4096, help="virtual MPI ranks = ILU(0) blocks of F, quoted for the 1.09 M-DoF mesh (larger meshes keep the rows per block)")
    ap.add_argument("--numbering", choices=("first_touch", "preordered"), default="first_touch",
                    help="what the caller hands over: first_touch = deal.II's own numbering on --ranks-input ranks per GPU, the virtual ranks and the "
                         "colour order are built inside libnsx (nsx_set_internal_layout); preordered = the front-end's virtual-rank numbering (rounds 1-3)")
    ap.add_argument("--ranks-input", type=int, default=1, help="ranks of the caller per GPU with --numbering first_touch (1 = the serial reference)")
    ap.add_argument("--schur-blocks", type=int, default=0,
                    help="ILU(0) blocks of the Schur matrix (--numbering preordered); 0 (default): consecutive ranks merged up to %d pressure rows per block" % SCHUR_ROWS)
    ap.add_argument("--ordering", choices=("colour", "first_touch", "colour_all"), default="colour",
                    help="node order inside a virtual rank")
    ap.add_argument("--balance", choices=("cells", "owned"), default="cells",
                    help="what the front-end's partitioner equalises over the ranks it is asked for: cells (METIS-like) or owned P2 nodes")
    ap.add_argument("--comm", choices=("auto", "rccl1"), default="auto",
                    help="rccl1 (one GPU): put a 1-rank RCCL communicator on the handle, i.e. time the DISTRIBUTED solver paths on one card (one launch + one "
                         "ncclAllReduce per reduction, the Gram-Schmidt sweep with the collective inside its persistent grid, the Schur CG in two launches per "
                         "iteration; no ghost exchange: one rank has no neighbours).  The line also carries the same run with a real RCCL kernel "
                         "(self-addressed send / receive, NSX_EXT_SELF_P2P=2) in front of every all-reduce")
    ap.add_argument("--no-big", action="store_true", help="N > 1: skip the strong_10M leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-only", action="store_true", help="only the cpu_baseline leg (after a short GPU run that provides its state)")
    ap.add_argument("--layout-table", metavar="FILE", help="write the preconditioner-layout table to FILE and exit")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--pmc", choices=("auto", "off"), default="auto",
                    help="auto: on one GPU with the cpu_baseline leg, measure roofline.traffic in two child runs under rocprofv3 --pmc "
                         "(FETCH_SIZE, WRITE_SIZE; ~1 min each); off: quote the committed profile")
    args = ap.parse_args()
    if os.environ.get("NSX_BENCH_HANG_DUMP"):  # development: where is every rank after that many seconds?
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["NSX_BENCH_HANG_DUMP"]), exit=False)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.steps is None:
        args.steps = 100 if max(world, args.gpus) == 1 else 30
    if args.spinup is None:
        args.spinup = 20 if max(world, args.gpus) == 1 else 10
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start one process per GPU as a CHILD (nothing here has touched the GPU yet) and relay its line
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT", "29517"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    if args.gpus > 1 and world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE JSON line and nothing else: libraries that print there (RCCL writes a version banner when a communicator is
    # created) are sent to stderr for the rest of the run; the line itself goes out through the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    live_pmc, pmc_outcome = None, {"status": "skipped", "why": "not requested for this invocation"}
    if args.pmc == "auto" and world == 1 and not args.no_cpu and not args.cpu_only and not args.layout_table:
        tail = ["--ranks", str(args.ranks), "--schur-blocks", str(args.schur_blocks), "--ordering", args.ordering, "--balance", args.balance,
                "--numbering", args.numbering, "--ranks-input", str(args.ranks_input), "--comm", args.comm]
        live_pmc, pmc_outcome = pmc_live(tail + (["--level", str(args.level)] if args.level is not None else []))
    if "NSX_BENCH_DEVICE" in os.environ:  # development only: several ranks on one card
        local_rank = int(os.environ["NSX_BENCH_DEVICE"])
    import torch
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(os.environ.get("NSX_BENCH_PG", "nccl"))  # "gloo" only for the one-card rehearsal
        on_cpu = dist.get_backend() == "gloo"

        def barrier():
            dist.barrier()
            torch.cuda.synchronize()

        def max_over_ranks(v):
            tt = torch.tensor([v], dtype=torch.float64, device="cpu" if on_cpu else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
    else:
        def barrier():
            torch.cuda.synchronize()

        def max_over_ranks(v):
            return v

    if args.layout_table:
        layout_table(local_rank, args.layout_table)
        return

    def partitioned_run(level, steps, warmup, spinup, profile_steps, want_state=False, twin=False):
        """every rank runs its part; a failure anywhere ends the job with a non-zero exit code (the launcher tears the
        group down) — a GPU fault must be investigated, not converted into a throughput number"""
        n_dofs_guess = BASE_DOFS if level == BASE_LEVEL else (BIG_DOFS if (level == BIG_LEVEL and BIG_DOFS) else None)
        ranks = total_ranks(args, n_dofs_guess) if n_dofs_guess else args.ranks * world
        beat("level %d: building the mesh and the caller's DoF table" % level, every=0.0 if level > 10 else 30.0)
        mesh, dofs, tables = build_problem(level, ranks, world, args.ordering, args.balance, args.numbering, args.ranks_input)
        try:
            el, stats, table, prof_stats, state = gpu_run(dofs, tables, steps, warmup, args.schur_blocks, local_rank, profile_steps=profile_steps,
                                                          barrier=barrier, rank=rank, world=world, want_state=want_state, spinup=spinup,
                                                          layout=layout_of(args, dofs, world), comm_single=(args.comm == "rccl1" and world == 1),
                                                          hoisted_twin=twin)
        except Exception as e:  # noqa: BLE001
            print("bench.py: rank %d failed in the partitioned run: %s: %s" % (rank, type(e).__name__, e), file=sys.stderr, flush=True)
            os._exit(3)  # peers may be blocked inside a collective: leave at once and let the launcher end them
        return max_over_ranks(el), stats, table, prof_stats, state, dofs

    level = args.level if args.level is not None else BASE_LEVEL
    steps, warmup = (args.steps, args.warmup) if not args.cpu_only else (2, 1)
    # ---- the headline: ONE mesh (1 089 643 DoF unless --level says otherwise) on N GPUs, plain time-steps/s
    elapsed, stats, table, prof_stats, state, dofs = partitioned_run(level, steps, warmup, args.spinup, 0 if args.cpu_only else args.profile_steps,
                                                                     want_state=(world == 1 and not args.no_cpu), twin=(world == 1 and not args.cpu_only and args.level is None))
    value = steps / elapsed
    run_big = world > 1 and args.level is None and not args.no_big
    self_p2p = None
    if args.comm == "rccl1" and world == 1 and not args.cpu_only and "NSX_EXT_SELF_P2P" not in os.environ:
        # a 1-rank ncclAllReduce launches no kernel: the same run once more with a self-addressed ncclSend / ncclRecv pair -- a real launch of
        # RCCL's device kernel -- in front of EVERY all-reduce, those inside the persistent sweep included (csrc/nsx_comm.hip)
        os.environ["NSX_EXT_SELF_P2P"] = "2"
        try:
            n2 = min(steps, 20)
            el2, st2, _, _, _, _ = partitioned_run(level, n2, warmup, args.spinup, 0)
            o2 = sum(s_["outer_iterations"] for s_ in st2)
            self_p2p = {"what": "the same run with NSX_EXT_SELF_P2P=2: a real RCCL kernel (self-addressed send / receive) in front of every all-reduce",
                        "steps": n2, "value": n2 / el2, "ms_per_step": 1e3 * el2 / n2, "gmres_outer_iters_per_step": o2 / float(n2),
                        "ms_per_outer_iteration": 1e3 * el2 / max(1, o2), "allreduces_per_step": st2[0].get("allreduces_per_step"),
                        "persistent_fallbacks": max(s_.get("persistent_fallbacks", 0) for s_ in st2), "paths": st2[0].get("paths")}
        finally:
            os.environ.pop("NSX_EXT_SELF_P2P", None)

    def big_leg(on_deadline):
        """strong_10M: the 10 644 763-DoF mesh partitioned over the N GPUs, the schedule of its committed one-GPU base.  The headline of
        this invocation is already measured when this leg starts: a rank that fails here, or a leg that outlives its deadline
        (NSX_BENCH_BIG_DEADLINE seconds), is REPORTED under "strong_10M" instead of taking the measured line with it"""
        import threading
        deadline = float(os.environ.get("NSX_BENCH_BIG_DEADLINE", "900"))
        finished = threading.Event()
        failure = []

        def watchdog():
            if not finished.wait(deadline):
                # the measured headline goes out first (rank 0); the other ranks leave a little later, so that the launcher does not end rank 0
                # before it has written.  A rank that RAISED inside the leg (an NSX_ERR_HIP from a GPU fault, a refused argument) makes the
                # job exit with code 4: the run record must show it.  A leg that is merely not finished at the deadline is reported in the
                # line and leaves with 0.
                if rank != 0:
                    time.sleep(5.0)
                on_deadline({"error": failure[0] if failure else "not finished after %.0f s (NSX_BENCH_BIG_DEADLINE)" % deadline, "n_gpus": world,
                             "failed": bool(failure)})
                os._exit(4 if failure else 0)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            n_ranks = total_ranks(args, BIG_DOFS) if BIG_DOFS else args.ranks * world
            beat("level %d: building the mesh and the caller's DoF table" % BIG_LEVEL, every=0.0)
            _, dofs_b, tables_b = build_problem(BIG_LEVEL, n_ranks, world, args.ordering, args.balance, args.numbering, args.ranks_input)
            el_b, stats_b, _, _, _ = gpu_run(dofs_b, tables_b, BIG_STEPS, BIG_WARMUP, args.schur_blocks, local_rank, profile_steps=0, barrier=barrier,
                                             rank=rank, world=world, spinup=BIG_SPINUP, layout=layout_of(args, dofs_b, world))
            el_b = max_over_ranks(el_b)
        except Exception as e:  # noqa: BLE001
            failure.append("rank %d: %s: %s" % (rank, type(e).__name__, e))
            print("bench.py: strong_10M leg failed on " + failure[0], file=sys.stderr, flush=True)
            # the peers are blocked inside a collective and an exit code here would make the launcher end rank 0 before it has written the
            # measured line: wait for the deadline, at which every rank leaves by itself
            while True:
                time.sleep(1.0)
        finished.set()
        outer_b = sum(s["outer_iterations"] for s in stats_b)
        big = {"time_steps_per_s": BIG_STEPS / el_b, "ms_per_step": 1e3 * el_b / BIG_STEPS, "n_dofs": dofs_b.n_dofs, "n_gpus": world,
               "steps": BIG_STEPS, "warmup": BIG_WARMUP, "spinup_steps": BIG_SPINUP,
               "gmres_outer_iters_per_step": outer_b / float(BIG_STEPS), "ms_per_outer_iteration": 1e3 * el_b / max(1, outer_b),
               "allreduces_per_step": stats_b[0].get("allreduces_per_step"), "ghost_exchanges_per_step": stats_b[0].get("ghost_exchanges_per_step"),
               "persistent_fallbacks": max(s.get("persistent_fallbacks", 0) for s in stats_b),
               "dof_steps_per_s_in_units_of_the_1M_mesh": BIG_STEPS / el_b * dofs_b.n_dofs / BASE_DOFS}
        base = committed_big_base()
        if base and base.get("n_dofs") == dofs_b.n_dofs and (base.get("steps"), base.get("warmup"), base.get("spinup_steps")) == (BIG_STEPS, BIG_WARMUP, BIG_SPINUP):  # (not in a rehearsal of this leg on another mesh / schedule)
            big["one_gpu_base"] = base
            big["speedup_over_one_gpu"] = big["time_steps_per_s"] / base["value"] if base.get("value") else None
            big["speedup_per_outer_iteration"] = base["ms_per_outer_iteration"] / big["ms_per_outer_iteration"] if base.get("ms_per_outer_iteration") else None
        return big

    if rank != 0:
        if run_big:
            big_leg(lambda err: None)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    if args.cpu_only:
        emit(cpu_baseline(dofs, state, local_rank))
        return

    outer = sum(s["outer_iterations"] for s in stats)
    t_solve = sum(s["t_solve"] for s in stats)
    kernels, kernel_ms_per_profiled_step = summarise_kernels(table, prof_stats)
    # roofline of the dominant kernel (by summed HIP-event time over the profiled steps)
    dom = max((k for k in kernels if kernels[k]["alg_GBps"]), key=lambda k: kernels[k]["share"]) if kernels else None
    roof = None
    if dom:
        a = kernels[dom]["alg_GBps"]
        traffic, traffic_src = pmc_traffic(dom, live_pmc) if dom in KERNEL_OF else (None, None)
        roof = {"kernel": dom, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                "traffic": traffic, "wasted": (traffic / table[dom]["bytes_per_launch"]) if traffic else None,
                "traffic_source": ("measured in this invocation: child runs of this command under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), "
                                   "average over the kernel's launches" if traffic_src == "live" else
                                   "constant from the committed rocprofv3 --pmc profile %s, not measured in this run" % traffic_src) if traffic_src else None,
                "algorithmic_bytes": table[dom]["bytes_per_launch"], "avg_us": kernels[dom]["avg_us"], "share_of_kernel_time": kernels[dom]["share"]}
        if "spmv_F" in kernels and kernels["spmv_F"]["alg_GBps"]:
            roof["spmv_F_GBps"] = kernels["spmv_F"]["alg_GBps"]
            roof["spmv_F_frac"] = kernels["spmv_F"]["alg_GBps"] / HBM_PEAK_GBS
    if live_pmc:
        for k, v in kernels.items():
            t, src = pmc_traffic(k, live_pmc) if k in KERNEL_OF else (None, None)
            if src == "live":
                v["pmc_bytes_per_launch"] = t
                v["wasted"] = t / table[k]["bytes_per_launch"] if table[k]["bytes_per_launch"] > 0 else None
    n = max(1, len(stats))
    lay = stats[0].get("layout") if stats else None
    caller = ("deal.II's own numbering handed over (first touch, %d rank(s) per GPU); %d virtual ranks, %s order and Schur blocks of <= %d rows built inside "
              "libnsx (nsx_set_internal_layout)" % (args.ranks_input, lay["ranks"] if lay else args.ranks, args.ordering, SCHUR_ROWS)) if args.numbering == "first_touch" else \
             ("pre-ordered numbering handed over: %d ranks (bisection balanced on %s; Schur: %s), %s node order inside a rank"
              % (total_ranks(args, dofs.n_dofs), "owned nodes" if args.balance == "owned" else "cells",
                 ("%d blocks" % args.schur_blocks) if args.schur_blocks else ("consecutive ranks merged up to %d rows per block" % SCHUR_ROWS), args.ordering))
    ms_outer = 1e3 * elapsed / max(1, outer)
    mean_long = long_run_mean_outer()
    kept = stats[0].get("hoisted") if stats else None
    out = {
        "metric": "time-steps/sec (assemble_time_step + solve_time_step), 3D flow past a cylinder, P2/P1, Yosida, per-rank ILU(0)",
        "value": value, "unit": "time-steps/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "spinup_steps": args.spinup, "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic (block-structured tetrahedral cylinder mesh, u0 = 0, reference inlet profile)",
        "config": {"workload": "3D flow-past-cylinder, P2/P1 (reference FE_SimplexP), %d DoF, %d cells, dt=2e-4, nu=1e-3, u_m=9, "
                               "GMRES(1e-4 abs)+Yosida(inner 1e-2), ILU(0) per rank; %s" % (dofs.n_dofs, dofs.n_cells, caller),
                   "n_dofs": dofs.n_dofs, "n_cells": dofs.n_cells, "numbering": args.numbering, "ranks_input_per_gpu": args.ranks_input if args.numbering == "first_touch" else None,
                   "internal_layout": lay,
                   "parallelism": (("1 GPU, no communication" if args.comm != "rccl1" else
                                    "1 GPU with a 1-rank RCCL communicator: the DISTRIBUTED solver paths (one launch + ncclAllReduce per reduction, collective inside the persistent sweep, two-launch Schur CG; no ghost exchange)")
                                   if world == 1 else
                                   "ONE mesh partitioned over %d GPUs: %s" % (world, "RCCL ghost exchange (grouped ncclSend/ncclRecv) + ncclAllReduce of the dot products"
                                                                          if os.environ.get("NSX_BENCH_COMM", "rccl") == "rccl" else
                                                                          "HOST CALLBACKS over torch.distributed/%s (development rehearsal, not RCCL)" % os.environ.get("NSX_BENCH_PG", "nccl")))},
        "gmres_outer_iters_per_step": outer / n,
        "ms_per_outer_iteration": ms_outer,
        "outer_iters_of_each_timed_step": [s["outer_iterations"] for s in stats],
        # the timed window is a sample of a chaotic sequence (restart steps come in runs): the committed long run's mean iteration count
        # times this run's cost per iteration is the rate a long simulation sees
        "value_long_run": (1e3 / (mean_long * ms_outer)) if mean_long else None,
        "long_run_mean_outer_iters_per_step": mean_long, "long_run_source": STEP_HISTORY if mean_long else None,
        "gmres_outer_iters_per_sec": outer / t_solve if t_solve > 0 else None,
        "inner_F_iters_per_step": sum(s["inner_F_iterations"] for s in stats) / n,
        "inner_S_iters_per_step": sum(s["inner_S_iterations"] for s in stats) / n,
        "setup_and_first_step_s": stats[0].get("setup_and_first_step_s") if stats else None,
        "t_prec_ms_per_step": 1e3 * sum(s["t_prec"] for s in stats) / n,
        # `value` is like for like: the timed steps rebuild the Schur product / ILU(S) / block inverses in every step as the reference does
        # (Preconditioners.hpp:358-362) unless NSX_SCHUR_CACHE was set from outside
        "schur_cache": stats[0].get("schur_cache") if stats else None,
        # the hoisted twin: the SAME timed steps run again from the restored state with those products kept while their inputs are bit-identical (Yosida: always)
        "value_hoisted": (steps / kept["elapsed"]) if kept else None,
        "hoisted": ({"value": steps / kept["elapsed"], "ms_per_step": 1e3 * kept["elapsed"] / steps, "t_prec_ms_per_step": 1e3 * kept["t_prec"],
                     "same_iteration_history": kept["outer"] == [s["outer_iterations"] for s in stats],
                     "what": "measured: the same %d timed steps from the restored state with NSX_SCHUR_CACHE=1 (Schur product, ILU(S) and block inverses kept across steps: their inputs do not change in time with Yosida)" % steps}
                    if kept else None),
        "paths": stats[0].get("paths") if stats else None,
        "self_p2p": self_p2p,
        "persistent_fallbacks": max(s.get("persistent_fallbacks", 0) for s in stats) if stats else None,
        "persistent_state": stats[0].get("persistent_state") if stats else None,
        "allreduces_per_step": stats[0].get("allreduces_per_step") if stats else None,
        "ghost_exchanges_per_step": stats[0].get("ghost_exchanges_per_step") if stats else None,
        "roofline": roof,
        "pmc_live": pmc_outcome,
        "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in kernels.items()},
        "kernel_profile": {"steps": len(prof_stats), "outer_iters_per_step": sum(s["outer_iterations"] for s in prof_stats) / max(1, len(prof_stats)),
                           "kernel_ms_per_step": kernel_ms_per_profiled_step,
                           "note": "per-kernel HIP-event pass over separate steps (not the timed ones); compare kernel_ms_per_step with "
                                   "ms_per_step x outer_iters_per_step / gmres_outer_iters_per_step"},
    }
    if world > 1:
        out["strong_1M"] = {"time_steps_per_s": value, "n_dofs": dofs.n_dofs, "note": "= value: the mesh of the N = 1 line on %d GPUs" % world}
    if out["persistent_fallbacks"] and world == 1:  # (N > 1: reported in the line — the launch-per-operation path is a legitimate, slower, distributed path)
        sys.exit("bench.py: %d persistent kernel(s) timed out and fell back to the launch-per-operation path: this is not the measured configuration"
                 % out["persistent_fallbacks"])
    layouts = committed_layouts()
    if layouts:
        out["preconditioner_layouts"] = layouts
    if run_big:
        def headline_only(err):
            out["strong_10M"] = err
            emit(out)
        out["strong_10M"] = big_leg(headline_only)
    if world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(dofs, state, local_rank)
        # lead with the stronger CPU code (compact storage, same algorithm): the stated baseline is the reference-shaped one
        out["gpu_over_best_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["best_cpu"]["value"]
        out["gpu_over_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["value"]
    emit(out)
    if os.environ.get("NSX_DUMP_MAPS"):
        # development aid (the round-2 abort inside exit() under rocprofv3): this process' mappings and the address of
        # libc's exit().  Library mappings keep their relative distances from run to run of one command in one image, so the
        # frames of a recorded stack can be translated by the difference of the two exit() addresses and looked up here.
        import ctypes
        with open(os.environ["NSX_DUMP_MAPS"], "w") as f:
            f.write("exit %#x\n" % ctypes.cast(ctypes.CDLL(None).exit, ctypes.c_void_p).value)
            f.write(open("/proc/self/maps").read())
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
