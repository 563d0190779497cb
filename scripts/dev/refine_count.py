"""developer check (GPU box): how many particles of the bench workload the finalize kernel hands to refine_winner_kernel"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from cryo_ralib_amd import api, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "mref"
cfg, nx, ou, xr, nref, _, _, _ = bench.WORKLOADS[wl]
n = 20000
dev = torch.device("cuda", 0)
refs_np = synth.make_references(max(nref, 1), nx, ou)
parts, _ = bench.generate_shard(dev, refs_np, n, xr, xr, 1.0, 0, nx, ou)
from cryo_ralib_amd.mref import MrefAligner, RefFreeAligner
if wl == "reffree":
    al = RefFreeAligner(parts, ou, xr, xr, 1.0, preprocess=True)
    step = lambda: al.iterate(-1, "ref_ali2d")
else:
    al = MrefAligner(parts, refs_np, ou, xr, xr, 1.0, preprocess=True)
    step = lambda: al.iterate("ref_ali2d", 1)
for it in range(3):
    step()
    torch.cuda.synchronize()
    print("iteration", it, "re-evaluated", al.engine.last_refine_count(), "of", n)
