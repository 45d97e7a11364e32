#!/usr/bin/env python3
"""Per-kernel HBM traffic from rocprofv3 --pmc passes.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_pmc_fetch_write_per_kernel.json

One counter per pass (the guide's HBM section: FETCH_SIZE and WRITE_SIZE cannot share a pass reliably).  Values are the
raw counter units (KiB) averaged over the launches of each kernel; bench.py applies the gfx950 correction
(FETCH_SIZE counts 64 B per 128-B request, so HBM read bytes = 2 * FETCH_SIZE KiB * 1024).
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    """Kernel symbol without its argument list (template arguments kept)."""
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def collect(root):
    acc = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                c = row["Counter_Name"]
                e = acc.setdefault(k, {}).setdefault(c, [0.0, set()])
                e[0] += float(row["Counter_Value"])
                e[1].add(row.get("Dispatch_Id") or row.get("Correlation_Id"))
    return acc


def main():
    out = {}
    for root in sys.argv[1:]:
        for k, counters in collect(root).items():
            if not re.search(r"nsx::", k):
                continue
            for c, (total, ids) in counters.items():
                e = out.setdefault(k, {})
                e["launches"] = len(ids)
                e[c + "_KB_avg"] = total / max(1, len(ids))
    json.dump(dict(sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_KB_avg", 0) * kv[1]["launches"])), sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
