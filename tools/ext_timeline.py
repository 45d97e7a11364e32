#!/usr/bin/env python3
"""Where a distributed Gram-Schmidt sweep with the collective inside its grid exchange spends its time: from a rocprofv3
--kernel-trace of `bench.py --comm rccl1`, per sweep the start / end of the persistent grid (k_mgs_one<.., true>), of k_ext_wait, of
whatever RCCL launches and of k_ext_release on the communication stream, relative to the grid's start.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 2 --warmup 1 --spinup 2 --no-cpu --profile-steps 0 --pmc off --comm rccl1
    python tools/ext_timeline.py gpurun_out/trace > profiles/rNN_ext_collective_timeline.txt"""
import csv
import glob
import os
import sys

import numpy as np


def main():
    rows = []
    for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    sweeps = [i for i, r in enumerate(rows) if "k_mgs_one" in r[2]]
    names = {}
    rec = []
    for i in sweeps:
        s0, e0 = rows[i][0], rows[i][1]
        # kernels of OTHER queues that end while the grid runs (k_ext_wait usually starts before it: the communication queue is idle)
        inside = [r for r in rows[max(0, i - 40):i + 8] if r[1] >= s0 and r[1] <= e0 + 1000 and r is not rows[i] and r[3] != rows[i][3]]
        d = {"grid": (0.0, (e0 - s0) / 1e3)}
        for r in inside:
            key = "k_ext_wait" if "k_ext_wait" in r[2] else "k_ext_release" if "k_ext_release" in r[2] else "other: " + r[2][:60]
            names[key] = names.get(key, 0) + 1
            d[key] = ((r[0] - s0) / 1e3, (r[1] - s0) / 1e3)
        rec.append(d)
    print("# %d sweeps; queues: grid %s" % (len(rec), rows[sweeps[0]][3] if sweeps else "-"))
    for key in ["grid"] + sorted(names):
        st = np.array([d[key][0] for d in rec if key in d])
        en = np.array([d[key][1] for d in rec if key in d])
        if len(st):
            print("%-70s n %6d  start us median %7.1f (p10 %7.1f p90 %7.1f)   end us median %7.1f (p10 %7.1f p90 %7.1f)"
                  % (key, len(st), np.median(st), np.percentile(st, 10), np.percentile(st, 90), np.median(en), np.percentile(en, 10), np.percentile(en, 90)))


if __name__ == "__main__":
    main()
