import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from cryo_ralib_amd import synth
from cryo_ralib_amd.mref import MrefAligner
nx, ou, xr, nref, n = 128, 60, 3, 10, 16384
dev = torch.device("cuda", 0)
refs = synth.make_references(nref, nx, ou)
parts, _ = bench.generate_shard(dev, refs, n, xr, xr, 1.0, 0, nx, ou)
al = MrefAligner(parts, refs, ou, xr, xr, 1.0, device=0, preprocess=True)
def t(f, name, reps=3):
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
        print("%-28s %.2f ms" % (name, (time.perf_counter() - t0) * 1e3))
t(lambda: al.iterate("ref_ali2d", 1), "iterate")
t(lambda: bench.live_offsets(al.state, nx, ou, xr, xr, 1.0, False), "live_offsets")
t(lambda: al.search(), "search")
gs = torch.zeros((nref, 2, nx, nx), device=dev); gc = torch.zeros(nref, dtype=torch.int32, device=dev)
t(lambda: al.engine.transform_accumulate(al.particles, al.result, 0, None, gs, gc), "transform_accumulate")
