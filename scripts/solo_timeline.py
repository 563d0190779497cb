"""prints the wave timeline recorded by scripts/solo_timeline.sh: per wave, the mean time per pass in each phase of
search_solo_kernel and at the barrier / counter behind it (clock64 ticks)"""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(64, 16, 16).astype(np.int64)
names = ["ring jobs", "wait ifft", "wait b1", "slice", "contract", "wait bA", "store", "wait bB", "ifft", "tail"]
passes = [g for g in range(64) if t[g, :, 0].all() and t[g, :, 9].all()]
print("passes recorded:", len(passes))
tot = np.zeros((16, len(names)))
for g in passes:
    d = np.diff(t[g, :, :10], axis=1)          # [wave][9 intervals between stamps 0..9]
    w5 = np.where(t[g, :, 13] > 0, t[g, :, 14] - t[g, :, 13], 0)
    tot[:, 0] += d[:, 0] - w5
    tot[:, 1] += w5
    tot[:, 2:] += d[:, 1:]
avg = tot / max(len(passes), 1)
print("wave " + " ".join("%10s" % n for n in names) + "   pass")
for w in range(16):
    print("%4d " % w + " ".join("%10.0f" % v for v in avg[w]) + "   %6.0f" % avg[w].sum())
print("mean " + " ".join("%10.0f" % v for v in avg.mean(0)) + "   %6.0f" % avg.sum(1).mean())
print("max  " + " ".join("%10.0f" % v for v in avg.max(0)))
samp = [(t[g, :, 13] - t[g, :, 0]) for g in passes if t[g, :, 13].all()]
if samp:
    print("sampling part of the ring job (start of pass -> counter wait inside the job), mean per wave:")
    print("     " + " ".join("%6.0f" % v for v in np.mean(samp, axis=0)))
if len(passes) > 1:
    st = np.array([t[g, :, 0].min() for g in passes])
    print("pass start -> next pass start (first wave), mean: %.0f ticks over %d passes; min %.0f max %.0f" % (np.diff(st).mean(), len(st) - 1, np.diff(st).min(), np.diff(st).max()))
