#!/bin/bash
# development tool (round 4), on the GPU box: per-workgroup wall-clock stamps of ONE call of the LDS-staged SpMV (k_spmv_blocked) in
# the middle of a bench run, launched five times: as it is (mode 0), without the x gathers (1), with the gathers from consecutive
# addresses (2), without the matrix stream (3), without any staging (4) -> gpurun_out/spmv_trace_mode<k>.txt, summarised by tools/spmv_trace.py
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-parameter"
make -B -C $R device HIPFLAGS="$BASE -DNSX_SPMV_TRACE" > $OUT/spmv_trace_build.log 2>&1 || exit 1
NSX_SPMV_TRACE_OUT=$OUT/spmv_trace NSX_SPMV_TRACE_CALL=${1:-6000} python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --pmc off --profile-steps 0 \
    > $OUT/spmv_trace.json 2> $OUT/spmv_trace.err || exit 2
for M in 0 1 2 3 4; do
  python3 $R/tools/spmv_trace.py $OUT/spmv_trace_mode$M.txt > $OUT/spmv_trace_mode$M.summary.txt || exit 3
done
make -B -C $R device > $OUT/spmv_trace_build_default.log 2>&1
