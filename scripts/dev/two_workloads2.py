import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from cryo_ralib_amd import synth
from cryo_ralib_amd.mref import MrefAligner
dev = torch.device("cuda", 0)
def run(nx, ou, nref, n):
    refs = synth.make_references(nref, nx, ou)
    parts, _ = bench.generate_shard(dev, refs, n, 3, 3, 1.0, 0, nx, ou)
    al = MrefAligner(parts, refs, ou, 3, 3, 1.0, device=0, preprocess=True)
    ts = []
    for i in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        al.iterate("ref_ali2d", 1)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(nx, ou, nref, n, " ".join("%.1f" % t for t in ts))
    al.close(); del al, parts; torch.cuda.empty_cache()
import sys
for a in sys.argv[1:]:
    run(*[int(v) for v in a.split(",")])
