#!/bin/bash
# development tool (GPU box): utilisation counters of the kernels of a short bench run, one counter group per pass
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "VALUBusy SALUBusy" "MemUnitBusy MemUnitStalled" "LDSBankConflict L2CacheHit" "MeanOccupancyPerCU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $R/bench.py --steps 1 --warmup 1 --spinup 1 --no-cpu --profile-steps 0 > $OUT/g$i.json 2> $OUT/g$i.err || echo "group $i failed"
  echo "group $i ($grp) done"
done
cd $R && python3 - <<'PY'
import csv, glob, os, collections, json
out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob("gpurun_out/pmc_probe/g*/**/*counter_collection.csv", recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0][:60]
            e = out[k][row["Counter_Name"]]
            e[0] += float(row["Counter_Value"]); e[1] += 1
res = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in out.items()}
json.dump(res, open("gpurun_out/pmc_probe/summary.json", "w"), indent=1)
for k in sorted(res, key=lambda k: -sum(out[k][c][1] for c in out[k]))[:8]:
    print(k, {c: round(v, 2) for c, v in res[k].items()})
PY
