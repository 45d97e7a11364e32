#!/bin/bash
# development tool (round 4), on the GPU box: does a REAL RCCL kernel run beside the persistent Gram-Schmidt grid?  (NSX_EXT_SELF_P2P=1: a
# self-addressed ncclSend / ncclRecv pair in front of every collective of the sweep, on a 1-rank communicator: RCCL's generic kernel,
# 256 threads x 264 VGPRs.)  Cases "<level> <NSX_MGS_MAXWG or -> <NSX_EXT_SELF_P2P>": bench line + whether the sweep fell back.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
i=0
while [ $# -ge 3 ]; do
  LEVEL=$1; MAXWG=$2; export NSX_EXT_SELF_P2P=$3; shift 3
  i=$((i+1))
  if [ "$MAXWG" = "-" ]; then unset NSX_MGS_MAXWG; else export NSX_MGS_MAXWG=$MAXWG; fi
  timeout -k 10 240 python3 $R/bench.py --comm rccl1 --level $LEVEL --steps 5 --warmup 2 --spinup 3 --no-cpu --profile-steps 2 --pmc off > $OUT/selfp2p_$i.json 2> $OUT/selfp2p_$i.err
  rc=$?
  echo "== level $LEVEL, NSX_MGS_MAXWG=$MAXWG, NSX_EXT_SELF_P2P=$NSX_EXT_SELF_P2P: rc $rc"
  grep -h "warning\|failed" $OUT/selfp2p_$i.err | head -3
  [ $rc -eq 0 ] && python3 - $OUT/selfp2p_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print("   %.3f steps/s, %.4f ms per outer iteration, fallbacks %s, %s; us: %s" % (d["value"], d["ms_per_outer_iteration"], d["persistent_fallbacks"], d["persistent_state"],
      {n: round(k[n]["avg_us"], 1) for n in ("mgs_sweep", "mgs_dots", "mgs_update", "spmv_F", "ilu_solve_F") if n in k}))
PY
done
