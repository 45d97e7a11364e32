#!/bin/bash
# round 3: Schur CG with the block inverses in registers (run on the GPU box through gpurun)
set -o pipefail
mkdir -p gpurun_out/r03
O=gpurun_out/r03
for cfg in "0 512" "1 512" "0 0" "1 0"; do
  set -- $cfg
  NSX_DEBUG=1 NSX_CG_PRES=$1 timeout -k 10 400 python bench.py --steps 20 --warmup 2 --spinup 10 --no-cpu --profile-steps 3 --schur-blocks $2 > $O/bench_cg_$1_$2.json 2> $O/bench_cg_$1_$2.err || { tail -20 $O/bench_cg_$1_$2.err; exit 1; }
  python - <<P
import json
d=json.load(open("$O/bench_cg_$1_$2.json"))
k=d["kernels"]
print("pres=$1 schur_blocks=$2", "ms/outer %.3f"%d["ms_per_outer_iteration"], "ms/step %.1f"%d["ms_per_step"], {n:round(k[n]["avg_us"],2) for n in ("ilu_solve_F","spmv_F","mgs_sweep","cg_S") if n in k}, "S its/step", d["inner_S_iters_per_step"], "outer/step", d["gmres_outer_iters_per_step"])
P
done
grep "persistent Schur" $O/bench_cg_1_0.err
timeout -k 10 600 python -m pytest tests/test_gpu_errors.py tests/test_gpu_parity.py -x -q -m gpu -k "schur or cg or dt0.001 or time_out" 2>&1 | tail -3
