#!/bin/bash
# development tool (round 4), on the GPU box: rocprofv3 kernel statistics of a short bench run for every value given of the switch $AB_VAR
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in ${@:-0 1}; do
  export ${AB_VAR:-NSX_SPMV_ORDER}=$V
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/order_stats_$V -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --profile-steps 0 --pmc off > $OUT/order_stats_$V.json 2> $OUT/order_stats_$V.err || exit 3
  find $OUT/order_stats_$V -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/order_kernel_stats_$V.csv
  rm -rf $OUT/order_stats_$V
  grep -E "k_spmv_blocked|k_ilu_solve_lanes|k_mgs_one" $OUT/order_kernel_stats_$V.csv | cut -c1-200
done
