"""per-pass spans of the fused-kernel wave timeline (scripts/fused_timeline.sh): which passes carry an image load"""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(64, 16, 16).astype(np.int64)
ok = [g for g in range(64) if t[g, :, 0].all()]
print("pass   start->start   ringjobs(mean)  sampling(mean)  first-wave-start  last-wave-start spread")
for a, b in zip(ok[:-1], ok[1:]):
    s0 = t[a, :, 0]; s1 = t[b, :, 0]
    rj = (t[a, :, 2] - t[a, :, 0]).mean() if t[a, :, 2].all() else -1
    sm = (t[a, :, 1] - t[a, :, 0]).mean() if t[a, :, 1].all() else -1
    print("%4d   %10d   %10.0f   %10.0f   %8d" % (a, s1.max() - s0.max(), rj, sm, s0.max() - s0.min()))
