#!/bin/bash
# Round profile set on the GPU box (run through gpurun from the repository root):
#   bench line with cpu_baseline, rocprofv3 kernel stats of the bench command, FETCH_SIZE / WRITE_SIZE passes.
# rocprofv3 gets the program itself after `--` (no wrapper), counters in runs of their own.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r04}
python3 $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --profile-steps 0 > $OUT/${TAG}_stats.json 2> $OUT/${TAG}_stats.err || exit 2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --profile-steps 0 > $OUT/${TAG}_pmc_fetch.json 2> $OUT/${TAG}_pmc_fetch.err || exit 3
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --profile-steps 0 > $OUT/${TAG}_pmc_write.json 2> $OUT/${TAG}_pmc_write.err || exit 4
cd $R && python3 tools/pmc_summary.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write > $OUT/${TAG}_pmc_fetch_write_per_kernel.json
find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_kernel_stats.csv
# the raw traces (one row per launch) are tens of MB: gpurun merges at most 64 MiB back, keep the summaries only
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
echo profile set done
