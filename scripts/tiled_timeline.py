"""prints the wave timeline of search_tiled_kernel recorded by a -DRALIGN_PROFILE_SWITCHES build with RALIGN_TIMELINE=<file>
(scripts/tiled_timeline.sh): per wave, mean clock ticks per pass spent in each phase of the first two reference tiles"""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(64, 16, 16).astype(np.int64)
passes = [g for g in range(64) if t[g, :, 0].all() and t[g, :, 15].all() and t[g, :, 13].all()]
print("passes recorded (with two full tiles):", len(passes))
names = ["ring jobs", "wait b1", "slice", "contr 0", "wait A0", "store 0", "wait B0", "ifft 0", "contr 1", "wait A1", "store 1", "wait B1", "ifft 1", "rest"]
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15]
tot = np.zeros((16, len(names)))
for g in passes:
    tot += np.diff(t[g][:, idx], axis=1)
avg = tot / max(len(passes), 1)
print("wave " + " ".join("%9s" % n for n in names) + "     pass")
for w in range(16):
    print("%4d " % w + " ".join("%9.0f" % v for v in avg[w]) + "   %7.0f" % avg[w].sum())
print("mean " + " ".join("%9.0f" % v for v in avg.mean(0)) + "   %7.0f" % avg.sum(1).mean())
print("max  " + " ".join("%9.0f" % v for v in avg.max(0)))
if passes:
    print("first pass start -> last recorded pass end: %d ticks (%d passes)" % (t[passes[-1], :, 15].max() - t[passes[0], :, 0].min(), len(passes)))
