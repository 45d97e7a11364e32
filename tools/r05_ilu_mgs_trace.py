"""development (round 5): summary of a k_ilu_mgs wave trace (NSX_ILU_MGS_TRACE=file): where does a launch spend its time?"""
import sys
import numpy as np
rows = [l.split() for l in open(sys.argv[1]) if not l.startswith("#")]
a = np.array([[int(v) for v in r] for r in rows], dtype=np.int64)
print(open(sys.argv[1]).readline().strip())
t = a[:, 3:].astype(float) / 100.0  # us
wave = a[:, 1]
names = ["start", "rhs in LDS", "fwd done", "bwd done", "basis arrived", "z complete", "local sums", "totals", "update done"]
def stat(x):
    x = x[x >= 0]
    return "min %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % (x.min(), np.median(x), np.percentile(x, 90), x.max()) if len(x) else "-"
for k, n in enumerate(names):
    print("%-14s all waves: %s" % (n, stat(t[:, k])))
sweep = t[:, 3] >= 0
print("sweeping waves: forward %s" % stat((t[sweep, 2] - t[sweep, 1])))
print("sweeping waves: backward %s" % stat((t[sweep, 3] - t[sweep, 2])))
print("sweeping waves: basis after sweeps %s" % stat((t[sweep, 4] - t[sweep, 3])))
print("other waves: basis arrived at %s" % stat(t[~sweep, 4]))
print("barrier released (z complete) - slowest sweep end per grid: %.2f ; last totals %.2f ; end %.2f" % (t[:, 5].max(), t[:, 7].max(), t[:, 8].max()))
