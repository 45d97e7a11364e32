#!/bin/bash
# round 3: ILU(0) factorisation with the rank block staged in LDS against the one through global memory
set -o pipefail
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_errors.py ${PYTEST_MORE} -x -q -m gpu > $O/pytest_ilu_factor.log 2>&1 || { tail -40 $O/pytest_ilu_factor.log; exit 1; }
tail -2 $O/pytest_ilu_factor.log
for v in ${VARIANTS:-1}; do
  NSX_ILU_FACTOR_LDS=$v timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu --pmc off > $O/bench_ilu_factor_$v.json 2> $O/bench_ilu_factor_$v.err || { tail -20 $O/bench_ilu_factor_$v.err; exit 1; }
  python - <<P
import json
d=json.load(open("$O/bench_ilu_factor_$v.json"))
k=d["kernels"]
print("lds=$v", "steps/s %.2f"%d["value"], "ms/outer %.3f"%d["ms_per_outer_iteration"], "outer/step %.1f"%d["gmres_outer_iters_per_step"], "ilu_factor_F", k["ilu_factor_F"], "t_prec %.3f cache-off %.3f"%(d["t_prec_ms_per_step"], d["t_prec_ms_per_step_cache_off"]))
P
done
