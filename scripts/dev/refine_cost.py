"""cost of the exact re-evaluation for EVERY particle (ra_set_refine(-1)) against the default threshold, headline geometry"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cryo_ralib_amd import api, synth
import bench
nx, ou, xr, nref, n = 90, 36, 3, 10, 50000
refs = synth.make_references(nref, nx, ou)
dev = torch.device("cuda:0")
tp, _ = bench.generate_shard(dev, refs, n, xr, xr, 1.0, 5, nx, ou)
for thr in (None, -1.0):
    eng = api.Engine(nx, ou, xr, xr, 1.0, nref, api.RA_MODE_MREF)
    if thr is not None: eng.set_refine(thr)
    eng.set_references(torch.from_numpy(refs).to(dev))
    st, res = eng.new_state(n), eng.new_result(n)
    for _ in range(2): eng.align(tp, st, res)
    eng.sync(); t0 = time.perf_counter()
    for _ in range(5): eng.align(tp, st, res)
    eng.sync(); dt = (time.perf_counter() - t0) / 5
    print("refine %s: %.3f ms per 50 000, refined %d" % (thr, dt * 1e3, eng.last_refine_count()))
    eng.close()
