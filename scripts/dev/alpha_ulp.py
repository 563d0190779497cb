"""particles whose refined alpha is not the checker's float: which step differs?  python scripts/dev/alpha_ulp.py nx ou ir rs xr ts nref n mode"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cryo_ralib_amd import api, synth
from oracle import oracle as orc
nx, ou, ir, rs, xr = (int(v) for v in sys.argv[1:6]); ts = float(sys.argv[6]); nref, n, mode = (int(v) for v in sys.argv[7:10])
refs = synth.make_references(nref, nx, ou)
parts, _ = synth.make_particles(refs, n, max(xr, 1), max(xr, 1), 0.4, ou=ou)
rg = orc.rings(ir, ou, rs); mask = orc.model_circle(ou, nx, nx)
refs_n, cref = orc.prepare_refs(refs, mask if mode == 0 else None, rg)
d = np.zeros((n, 2), np.float32)
if len(sys.argv) > 10:
    d = (np.random.default_rng(3).integers(-2, 3, size=(n, 2)) * ts).astype(np.float32)
d0 = d.copy()
if mode == 0:
    params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, ts, d, nthreads=16)
else:
    p0 = np.zeros((n, 6), np.float32); p0[:, 1:3] = -d0
    params, infos, _, _ = orc.reffree_iteration(parts, cref[0], rg, xr, xr, ts, (0, 0), d, p0, nthreads=16)
eng = api.Engine(nx, ou, xr, xr, ts, nref, mode, first_ring=ir, ring_skip=rs)
eng.set_refine(-1.0)
eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
st, res = torch.from_numpy(d0.copy()).to(eng.dev), eng.new_result(n)
eng.align(torch.from_numpy(parts).to(eng.dev), st, res); eng.sync()
r = api.Engine.result_to_numpy(res)
mis = np.where((r["alpha"] != params[:, 0]) | (r["sx"] != params[:, 1]) | (r["sy"] != params[:, 2]))[0]
print("%d of %d particles differ in alpha / sx / sy" % (len(mis), n))
f32 = np.float32
for i in mis[:12]:
    tot = f32(infos[i].tot)
    ang = np.fmod(((tot - f32(1.0)) / f32(rg.maxrin) + f32(1.0)) * f32(360.0), f32(360.0))
    print("particle %d: bin %d | alpha engine %.9g checker %.9g | sx %.9g %.9g | sy %.9g %.9g | checker tot %.9g -> ang by the engine's float formula %.9g" % (
        i, r["angle_bin"][i], r["alpha"][i], params[i, 0], r["sx"][i], params[i, 1], r["sy"][i], params[i, 2], tot, ang))
