"""developer measurement: BASELINE configs[2] -- reference-free ali2d, 50k synthetic 90x90 particles, ts=1, xr=yr=3,
ou=36, 10 iterations (one reference = the running average), on one GPU.  Prints particles/s per iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cryo_ralib_amd import synth
from cryo_ralib_amd.mref import RefFreeAligner
import bench

def main(n=50000, nx=90, ou=36, xr=3, iters=10):
    dev = torch.device("cuda", 0)
    refs = synth.make_references(1, nx, ou)
    parts, _ = bench.generate_shard(dev, refs, n, xr, xr, 1.0, 0, nx, ou)
    al = RefFreeAligner(parts, ou, xr, xr, 1.0, device=0)
    al.iterate(-1, "ref_ali2d")
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(iters):
        al.iterate(-1, "ref_ali2d")
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("reference-free: %d particles x %d iterations in %.3f s = %.0f particles/s; criterion %.4g -> %.4g" %
          (n, iters, dt, n * iters / dt, al.criteria[0], al.criteria[-1]))
    al.close()

if __name__ == "__main__":
    main()
