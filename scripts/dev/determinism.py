"""the same stack through the same engine three times and through a fresh engine: the records must be the same to the bit (a race
between waves -- an LDS hazard, a missing barrier -- shows as a record that differs from run to run).  Batches large enough to fill
the GPU several times over.  python scripts/dev/determinism.py"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cryo_ralib_amd import api, synth
import bench

GEOS = [(90, 36, 3, 10, 0, 1.0, 6000), (90, 36, 3, 1, 1, 1.0, 6000), (90, 36, 3, 50, 0, 1.0, 3000), (44, 10, 2, 3, 0, 1.0, 6000),
        (100, 40, 3, 10, 0, 1.0, 4000), (130, 52, 3, 50, 0, 1.0, 2000), (128, 60, 2, 4, 1, 0.5, 2000), (150, 66, 2, 3, 0, 1.0, 1500),
        (256, 36, 3, 10, 0, 1.0, 2000), (112, 30, 3, 24, 0, 1.0, 3000), (256, 120, 5, 100, 0, 1.0, 700), (200, 90, 3, 20, 0, 0.5, 600)]
for nx, ou, xr, nref, mode, ts, n in GEOS:
    refs = synth.make_references(nref, nx, ou)
    dev = torch.device("cuda:0")
    tp, _ = bench.generate_shard(dev, refs, n, xr, xr, 1.0, 5, nx, ou)
    d0 = (torch.randint(-2, 3, (n, 2), device=dev).float() * ts)
    outs = []
    for fresh in range(2):
        eng = api.Engine(nx, ou, xr, xr, ts, nref, mode)
        eng.set_references(torch.from_numpy(refs).to(dev))
        for rep in range(3 if fresh == 0 else 1):
            st, res = d0.clone(), eng.new_result(n)
            eng.align(tp, st, res)
            sums = torch.zeros((nref, 2, nx, nx), dtype=torch.float32, device=dev)
            counts = torch.zeros(nref, dtype=torch.int32, device=dev)
            eng.transform_accumulate(tp, res, 0, None, sums, counts)
            eng.sync()
            outs.append((res.clone(), st.clone(), sums.clone(), counts.clone()))
        path = eng.search_path
        eng.close()
    ok = all(torch.equal(outs[0][k], o[k]) for o in outs[1:] for k in range(4))
    ndiff = [int((outs[0][0] != o[0]).any(1).sum().item()) for o in outs[1:]]
    print("nx=%d ou=%d xr=%d nref=%d mode=%d ts=%g n=%d path %d: %s %s" % (nx, ou, xr, nref, mode, ts, n, path, "bitwise equal over 4 runs" if ok else "DIFFERENT", "" if ok else ndiff), flush=True)
    assert ok
print("done")
