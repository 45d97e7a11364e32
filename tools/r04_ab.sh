#!/bin/bash
# development tool (round 4), on the GPU box: A/B of two compile-time choices in ONE session (boxes differ by a few per cent):
#   NSX_ILU_STREAM_ALIGN  8 (rounds 1-3) / 4: padding of a sweep of the lane-owner triangular-solve stream
#   NSX_MGS_OLD_NT        1 (rounds 2-3) / 0: cache policy of the basis vectors a Gram-Schmidt sweep reads twice
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-parameter"
for V in "8 1" "4 1" "8 0" "4 0"; do
  set -- $V
  TAG=align$1_oldnt$2
  make -B -C $R device HIPFLAGS="$BASE -DNSX_ILU_STREAM_ALIGN=$1 -DNSX_MGS_OLD_NT=$2" > $OUT/r04_ab_build_$TAG.log 2>&1 || exit 1
  python3 $R/bench.py --steps 30 --warmup 5 --no-cpu --pmc off > $OUT/r04_ab_$TAG.json 2> $OUT/r04_ab_$TAG.err || exit 2
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_ab_stats_$TAG -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --profile-steps 0 --pmc off > $OUT/r04_ab_stats_$TAG.json 2> $OUT/r04_ab_stats_$TAG.err) || exit 3
  find $OUT/r04_ab_stats_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r04_ab_kernel_stats_$TAG.csv
  rm -rf $OUT/r04_ab_stats_$TAG
  echo "$TAG done"
done
make -B -C $R device > $OUT/r04_ab_build_default.log 2>&1
