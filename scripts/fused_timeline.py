"""prints the wave timeline recorded by scripts/fused_timeline.sh: per pass, the time every wave spent in each phase and
at the barrier behind it (clock64 ticks -> ns at 100 MHz if the values look like the constant clock, else raw cycles)"""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(64, 16, 16).astype(np.int64)
names = ["ring jobs", "wait ifft", "wait b1", "contraction", "wait b3", "store", "wait b4", "ifft", "wait b5"]
passes = [g for g in range(64) if t[g, :, 0].all()]
print("passes recorded:", len(passes))
t0 = t[passes[0], :, 0].min()
tot = np.zeros((16, 9))
for g in passes:
    d = np.diff(t[g, :, :9], axis=1)          # [wave][8 phases]
    # stamps 13 / 14: before / after the barrier that ends the previous pass's inverse FFTs, taken inside the ring job
    # (zero in the first pass, which has nothing to wait for)
    w5 = np.where(t[g, :, 13] > 0, t[g, :, 14] - t[g, :, 13], 0)
    tot[:, 0] += d[:, 0] - w5
    tot[:, 1] += w5
    tot[:, 2:] += d[:, 1:]
avg = tot / len(passes)
print("average ticks per pass, per wave (rows = waves 0..15); 'wait ifft' = at the barrier inside the ring job (after the sampling)")
print("that ends the previous pass's inverse FFTs, 'wait b5' = the same barrier after the last pass:")
print("wave " + " ".join("%12s" % n for n in names) + "   pass")
for w in range(16):
    print("%4d " % w + " ".join("%12.0f" % v for v in avg[w]) + "   %6.0f" % avg[w].sum())
print("mean " + " ".join("%12.0f" % v for v in avg.mean(0)) + "   %6.0f" % avg.sum(1).mean())
print("max  " + " ".join("%12.0f" % v for v in avg.max(0)))
samp = [(t[g, :, 13] - t[g, :, 0]) for g in passes if t[g, :, 13].all()]
if samp:
    print("sampling part of the ring job (start of pass -> barrier inside the job), mean per wave:")
    print("     " + " ".join("%6.0f" % v for v in np.mean(samp, axis=0)))
span = (t[passes[-1], :, 8].max() - t0)
print("first pass start -> last pass end: %d ticks (%d passes)" % (span, len(passes)))

# contraction loop: stamps 9.. = start of every second ring quad (rf_contract), relative to the barrier before it (stamp 2)
it = t[passes, :, 9:13] - t[passes, :, 2:3]
it = np.where(t[passes, :, 9:13] > 0, it, 0)
print("contraction: start of loop iteration i after barrier 1 (mean over passes), then end of the phase:")
for w in range(16):
    row = it[:, w, :].mean(0)
    print("%4d " % w + " ".join("%7.0f" % v for v in row) + "   end %7.0f" % (t[passes, w, 3] - t[passes, w, 2]).mean())
