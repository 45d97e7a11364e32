"""development tool: summary of the per-workgroup stamps tools/spmv_trace.sh records for one launch of k_spmv_blocked
(columns: block chunk rows ucols t_start t_staged t_end xcc hw_id; ticks of 10 ns)."""
import sys

import numpy as np


def main(path):
    a = np.loadtxt(path, dtype=np.int64, ndmin=2)
    blk, chunk, rows, ucols, t0, t1, t2, xcc, hw = a.T
    base = t0.min()
    t0, t1, t2 = (t0 - base) / 100.0, (t1 - base) / 100.0, (t2 - base) / 100.0  # us
    print("workgroups %d, launch span %.2f us (first start -> last end)" % (len(a), t2.max()))
    print("start times  : p0 %.2f p25 %.2f p50 %.2f p75 %.2f p100 %.2f us" % tuple(np.percentile(t0, [0, 25, 50, 75, 100])))
    print("end times    : p0 %.2f p25 %.2f p50 %.2f p75 %.2f p100 %.2f us" % tuple(np.percentile(t2, [0, 25, 50, 75, 100])))
    life, stage, stream = t2 - t0, t1 - t0, t2 - t1
    for name, v in (("lifetime", life), ("staging", stage), ("streaming", stream)):
        print("%-9s    : mean %.2f p5 %.2f p50 %.2f p95 %.2f max %.2f us" % ((name, v.mean()) + tuple(np.percentile(v, [5, 50, 95, 100]))))
    print("rows / workgroup: mean %.1f max %d; staged columns: mean %.1f max %d" % (rows.mean(), rows.max(), ucols.mean(), ucols.max()))
    # residency over time
    edges = np.arange(0.0, t2.max() + 1.0, 1.0)
    print("time [us] : workgroups resident / staging / streaming / started / ended in that microsecond")
    for lo in edges[:-1]:
        hi = lo + 1.0
        mid = lo + 0.5
        res = ((t0 <= mid) & (t2 > mid)).sum()
        stg = ((t0 <= mid) & (t1 > mid)).sum()
        print("  %5.1f   : %5d %5d %5d %5d %5d" % (lo, res, stg, res - stg, ((t0 >= lo) & (t0 < hi)).sum(), ((t2 >= lo) & (t2 < hi)).sum()))
    # per XCD and per CU balance
    cu = (xcc << 16) | ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 16) & 0x1) << 7)  # CU_ID [11:8], SE_ID [15:13], SA [16]
    n_per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
    print("distinct (xcc, se, sa, cu): %d; workgroups per CU: min %d mean %.1f max %d" % (len(n_per_cu), n_per_cu.min(), n_per_cu.mean(), n_per_cu.max()))
    for x in np.unique(xcc):
        m = xcc == x
        print("  xcc %d: %d workgroups, rows %d, last end %.2f us" % (x, m.sum(), rows[m].sum(), t2[m].max()))
    # correlation of lifetime with size
    print("corr(lifetime, rows) %.2f, corr(lifetime, staged columns) %.2f, corr(streaming, rows) %.2f" %
          (np.corrcoef(life, rows)[0, 1], np.corrcoef(life, ucols)[0, 1], np.corrcoef(stream, rows)[0, 1]))


if __name__ == "__main__":
    main(sys.argv[1])
