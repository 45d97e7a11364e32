#!/bin/bash
# development tool (round 4), on the GPU box: load-path counters of the LDS-staged SpMV (one rocprofv3 --pmc pass per counter group)
#   bash tools/r04_pmc_spmv.sh "TA_BUSY_avr GRBM_GUI_ACTIVE" "MemUnitBusy MemUnitStalled" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
# (results: profiles/r04_spmv_workgroup_timeline.txt).  The group "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
# TA_DATA_STALLED_BY_TC_CYCLES_sum" cannot be collected in ONE pass: rocprofv3 answers "Could not construct profile cfg ... error code 38:
# Request exceeds the capabilities of the hardware" (three TA counters in one pass), aborts (signal 6) and the process then never exits --
# round 4's run sat silent until it was ended after 7 minutes (gpurun_out/pmc_spmv_4.err).  A profiler limit, not a library hang: split
# such a group over several passes.  Every pass below runs under `timeout`, so a group the hardware refuses costs two minutes, not the call.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/pmc_list_avail.txt 2>&1 || true
i=0
for G in "$@"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/pmc_spmv_$i -- python3 $R/bench.py --steps 1 --warmup 1 --spinup 2 --no-cpu --profile-steps 0 --pmc off > $OUT/pmc_spmv_$i.json 2> $OUT/pmc_spmv_$i.err || { echo "group $i ($G) failed or was refused by the profiler (error code 38 = too many counters of one block in a pass)"; grep -m1 "error code" $OUT/pmc_spmv_$i.err; tail -3 $OUT/pmc_spmv_$i.err; break; }
  python3 - "$OUT/pmc_spmv_$i" "$G" <<'PY'
import csv, glob, sys, collections
d, names = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"]
    for key in ("k_spmv_blocked", "k_ilu_solve_lanes", "k_mgs_one<8", "k_axpy_multi", "k_spmv_vel"):
        if key in k:
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key, cs in acc.items():
    print(key, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in cs.items()})
PY
  rm -rf $OUT/pmc_spmv_$i
done
