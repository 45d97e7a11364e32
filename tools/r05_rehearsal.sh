#!/bin/bash
# development tool (round 5), on the one-GPU box: the driver's multi-GPU command, `bench.py --gpus N`, rehearsed with N processes on ONE
# card over the host-callback backend (torch.distributed / gloo instead of RCCL, which refuses two ranks on one device), headline mesh
# AND the 10 644 763-DoF strong_10M leg.  N <= 6: the pool allows six processes on a card.
#   bash tools/r05_rehearsal.sh 6 "3,1,1" out.json     (ranks, "steps,warmup,spinup" of the strong_10M leg, output)
# Every rank logs the paths it took and the ones it WOULD take under RCCL (nsx_path_info) to <output>.err.
set -o pipefail
N=${1:-6}
SCHED=${2:-3,1,1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${3:-$R/gpurun_out/r05_rehearsal_$N.json}
export NSX_BENCH_COMM=callbacks NSX_BENCH_PG=gloo NSX_BENCH_DEVICE=0 NSX_BENCH_BIG_SCHEDULE=$SCHED NSX_BENCH_BIG_DEADLINE=${NSX_BENCH_BIG_DEADLINE:-900}
export NSX_BENCH_HANG_DUMP=${NSX_BENCH_HANG_DUMP:-600} HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
cd $R
timeout -k 10 ${REHEARSAL_TIMEOUT:-1100} python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus $N --steps ${REHEARSAL_STEPS:-4} --warmup 1 --spinup 2 --no-cpu > $OUT 2> $OUT.err
rc=$?
echo "rehearsal rc=$rc" >> $OUT.err
grep -h "paths\|strong_10M\|failed\|Error\|warning" $OUT.err | cut -c1-600 | tail -40
exit $rc
