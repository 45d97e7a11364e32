#!/bin/bash
# round 3: the one-exchange Gram-Schmidt sweep (k_mgs_one) against two links per exchange (k_mgs_blk)
set -o pipefail
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_errors.py -x -q -m gpu -k "gram_schmidt or time_out or two_handles" 2>&1 | tail -3 || exit 1
for links in ${LINKS:-2 0}; do
  NSX_DEBUG=1 NSX_MGS_LINKS=$links timeout -k 10 400 python bench.py --steps 20 --warmup 2 --spinup 10 --no-cpu --profile-steps 3 > $O/bench_mgs_$links.json 2> $O/bench_mgs_$links.err || { tail -20 $O/bench_mgs_$links.err; exit 1; }
  python - <<P
import json
d=json.load(open("$O/bench_mgs_$links.json"))
k=d["kernels"]
print("links=$links", "ms/outer %.3f"%d["ms_per_outer_iteration"], "ms/step %.1f"%d["ms_per_step"], {n:round(k[n]["avg_us"],2) for n in ("ilu_solve_F","spmv_F","mgs_sweep","cg_S") if n in k}, "outer/step", d["gmres_outer_iters_per_step"], "F/step", d["inner_F_iters_per_step"], d["outer_iters_of_each_timed_step"])
P
done
