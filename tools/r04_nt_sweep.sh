#!/bin/bash
# development tool (round 4), on the GPU box: the cache-policy mask -DNSX_NT (nsx_internal.hpp: 1 = Krylov basis, 2 = SpMV matrix
# stream, 4 = triangular-solve factor stream non-temporal) re-measured with this round's kernels: rocprofv3 kernel statistics per mask
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-parameter"
for V in ${@:-1 4 5 0}; do
  make -B -C $R device HIPFLAGS="$BASE -DNSX_NT=$V" > $OUT/nt_build_$V.log 2>&1 || exit 1
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nt_stats_$V -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --profile-steps 0 --pmc off > $OUT/nt_stats_$V.json 2> $OUT/nt_stats_$V.err) || exit 3
  find $OUT/nt_stats_$V -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/nt_kernel_stats_$V.csv
  rm -rf $OUT/nt_stats_$V
  echo "== NSX_NT=$V"
  python3 - $OUT/nt_kernel_stats_$V.csv $OUT/nt_stats_$V.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:5]:
    print("  %-46s %7s calls  avg %8.2f us  %5.2f %%" % (r["Name"][:46], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("  total kernel time %.1f ms; bench: %.3f steps/s, %.4f ms per outer iteration" % (tot / 1e6, d["value"], d["ms_per_outer_iteration"]))
PY
done
make -B -C $R device > $OUT/nt_build_default.log 2>&1
