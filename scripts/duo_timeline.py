"""wave timeline of search_duo_kernel (scripts/solo_timeline.sh with RALIGN_DUO=1): mean ticks per pass between the stamps"""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(64, 16, 16).astype(np.int64)
names = ["ringA", "bar", "sliceA", "bar", "ringB", "bar", "sliceB", "contr", "barA", "store", "barB", "ifft", "tail"]
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 15]
passes = [g for g in range(64) if all(t[g, :, i].all() for i in idx)]
print("two-offset passes recorded:", len(passes))
tot = np.zeros((16, len(names)))
for g in passes:
    st = t[g][:, idx]
    tot += np.diff(st, axis=1)
avg = tot / max(len(passes), 1)
print("wave " + " ".join("%7s" % n for n in names) + "   pass")
for w in range(16):
    print("%4d " % w + " ".join("%7.0f" % v for v in avg[w]) + "   %6.0f" % avg[w].sum())
print("mean " + " ".join("%7.0f" % v for v in avg.mean(0)) + "   %6.0f" % avg.sum(1).mean())
print("max  " + " ".join("%7.0f" % v for v in avg.max(0)))
