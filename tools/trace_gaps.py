#!/usr/bin/env python3
"""Idle time between consecutive kernels of one rocprofv3 --kernel-trace run, grouped by (kernel before, kernel after).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu --profile-steps 0 --pmc off
    python tools/trace_gaps.py gpurun_out/trace > profiles/rNN_kernel_gaps.txt

Only the launches between the first and the last nsx::k_mgs* launch are looked at (the time steps, not the set-up)."""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import short


def main():
    rows = []
    for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]).replace("void ", "").replace("nsx::", "")))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if r[2].startswith("k_mgs")]
    rows = rows[idx[0]:idx[-1] + 1]
    busy = sum(e - s for s, e, _ in rows)
    wall = rows[-1][1] - rows[0][0]
    pair = defaultdict(lambda: [0, 0.0, []])
    for a, b in zip(rows, rows[1:]):
        g = max(0, b[0] - a[1])
        e = pair[(a[2], b[2])]
        e[0] += 1
        e[1] += g
        e[2].append(g)
    print("# %d launches, wall %.2f ms, kernels busy %.2f ms (%.1f %%), idle %.2f ms" % (len(rows), wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6))
    print("# %-34s -> %-34s %7s %9s %9s %9s %8s" % ("kernel before", "kernel after", "count", "idle ms", "median us", "p90 us", "share %"))
    tot = sum(v[1] for v in pair.values()) or 1.0
    for (a, b), (n, t, gs) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:24]:
        gs.sort()
        print("  %-34s -> %-34s %7d %9.3f %9.2f %9.2f %8.1f" % (a[:34], b[:34], n, t / 1e6, gs[len(gs) // 2] / 1e3, gs[int(0.9 * (len(gs) - 1))] / 1e3, 100.0 * t / tot))


if __name__ == "__main__":
    main()
