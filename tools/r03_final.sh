#!/bin/bash
# round 3: the committed measurement set (run on the GPU box through gpurun): profile set, driver-shaped bench line, microbench, Schaefer-Turek
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
bash tools/profile_round.sh r03 || exit 1
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_bench_driver_shape.json 2> $O/r03_bench_driver_shape.err || exit 2
timeout -k 5 200 tools/ilu_lanes_bench 208 > $O/r03_lanes_bench.txt 2>&1 || exit 3
timeout -k 10 900 python3 tools/schaefer_turek_2d.py --levels 4 8 12 > $O/r03_schaefer_turek_2d.txt 2> $O/r03_st.err || exit 4
echo final set done
