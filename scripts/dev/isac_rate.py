"""throughput of the class-resident alignment (ref_free_alignment_2D): ncls classes of m particles each"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cryo_ralib_amd import api, synth
nx, ou, xr = 90, 36, 3
ncls, m = int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 100
refs = synth.make_references(min(ncls, 20), nx, ou)
refs = np.concatenate([refs] * ((ncls + len(refs) - 1) // len(refs)))[:ncls]
p, _ = synth.make_particles(refs[:1], m, xr, xr, 0.5, ou=ou)
parts = np.concatenate([p] * ncls)
n = ncls * m
cid = (ctypes.c_int * n)(*[i // m for i in range(n)])
lib = api.load_library()
cfg = api.AlignConfig(n, ncls, nx, ou, 256, 1.0, float(xr), float(xr))
prm = lib.ref_free_alignment_2D_init(ctypes.byref(cfg), api.get_c_ptr_array(list(parts)), api.get_c_ptr_array(list(refs)), cid, 0)
lib.ref_free_alignment_2D()
t0 = time.perf_counter()
its = 3
for _ in range(its):
    lib.ref_free_alignment_2D()
dt = (time.perf_counter() - t0) / its
print("classes %d x %d particles: %.1f ms per iteration = %.0f particles/s" % (ncls, m, dt * 1e3, n / dt))
lib.gpu_clear()
