"""alignment parameters of the large-box configuration (256 x 256, ou = 120, 100 references) against the oracle:
prints the largest angle / shift deviations (sub-bin interpolation is ill conditioned on flat peaks)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_gpu_parity import synth, oracle_setup, run_engine, orc, api
nx, ou, nref, xr, n = 256, 120, 100, 5, int(sys.argv[1]) if len(sys.argv) > 1 else 4
refs = synth.make_references(nref, nx, ou)
parts, truth = synth.make_particles(refs, n, xr, xr, 0.25, ou=ou)
rg, mask, refs_n, cref = oracle_setup(refs, ou, nx)
d = np.zeros((n, 2), np.float32)
params, infos, sums, counts = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d, nthreads=16)
eng, tp, st, res = run_engine(parts, refs_n, ou, xr, xr, 1.0)
r = api.Engine.result_to_numpy(res)
p = np.asarray(params)
print("oracle params[0]:", p[0])
da = np.abs(((r["alpha"] - p[:, 0] + 180) % 360) - 180)
print("max |dalpha| deg", da.max(), " in bins of 360/1024:", da.max() / (360 / 1024))
print("max |dsx|, |dsy|", np.abs(r["sx"] - p[:, 1]).max(), np.abs(r["sy"] - p[:, 2]).max())
print("alpha gpu", r["alpha"], "oracle", p[:, 0])
print("mirror", r["mirror"], p[:, 3], "ref", r["ref_id"], p[:, 4] if p.shape[1] > 4 else "")
