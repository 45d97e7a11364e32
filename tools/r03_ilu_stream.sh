#!/bin/bash
# round 3: the lane-owner triangular solve against the round-2 lane-group stream (run on the GPU box through gpurun)
set -o pipefail
mkdir -p gpurun_out/r03
O=gpurun_out/r03
[ -n "$SKIP_TESTS" ] || timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_levelled.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest_a.log 2>&1 || { tail -30 $O/pytest_a.log; exit 1; }
[ -n "$SKIP_TESTS" ] || tail -3 $O/pytest_a.log
CFGS=${CFGS:-0:1:8 1:4:8 1:6:8 1:8:8 1:12:8 1:8:16 1:8:4}
for cfg in $CFGS; do
  IFS=: read st bpw pf ept <<< "$cfg"
  NSX_DEBUG=1 NSX_ILU_STREAM=$st NSX_BPW_F=$bpw NSX_PF=$pf NSX_ILU_EPT=${ept:-2} timeout -k 10 400 python bench.py --steps 10 --warmup 2 --spinup 5 --no-cpu --profile-steps 3 > $O/bench_${st}_${bpw}_${pf}_${ept}.json 2> $O/bench_${st}_${bpw}_${pf}_${ept}.err || { tail -20 $O/bench_${st}_${bpw}_${pf}_${ept}.err; exit 1; }
  python - <<P
import json
d=json.load(open("$O/bench_${st}_${bpw}_${pf}_${ept}.json"))
k=d["kernels"]
print("stream=$st bpw=$bpw pf=$pf ept=$ept", "ms/outer %.3f"%d["ms_per_outer_iteration"], {n:round(k[n]["avg_us"],2) for n in ("ilu_solve_F","spmv_F","mgs_sweep","cg_S") if n in k}, d["roofline"]["kernel"], round(d["roofline"]["frac"],3), d["outer_iters_of_each_timed_step"])
P
done
