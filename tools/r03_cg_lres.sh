#!/bin/bash
# round 3: Schur CG with the operator's rows resident in LDS against the slab stream
set -o pipefail
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_errors.py -x -q -m gpu -k "schur_cg or persistent" > $O/pytest_cg_lres.log 2>&1 || { tail -40 $O/pytest_cg_lres.log; exit 1; }
for v in 0 1; do
  NSX_CG_LRES=$v timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu --pmc off > $O/bench_cg_lres_$v.json 2> $O/bench_cg_lres_$v.err || { tail -20 $O/bench_cg_lres_$v.err; exit 1; }
  python - <<P
import json
d=json.load(open("$O/bench_cg_lres_$v.json"))
k=d["kernels"]
print("lres=$v", "steps/s %.2f"%d["value"], "ms/outer %.3f"%d["ms_per_outer_iteration"], "outer/step %.1f"%d["gmres_outer_iters_per_step"], "cg_S", k["cg_S"], d["inner_S_iters_per_step"], d["persistent_fallbacks"])
P
done
