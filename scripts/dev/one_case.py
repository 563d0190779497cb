"""one geometry through the engine against the CPU checker, with the per-particle figures printed:
python scripts/dev/one_case.py nx ou ir rs xr yr ts nref n mode(0|1) [state-seed | -1] [interp] [norm]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cryo_ralib_amd import api, synth          # noqa: E402
from oracle import oracle as orc               # noqa: E402

a = sys.argv[1:]
nx, ou, ir, rs, xr, yr = (int(v) for v in a[:6])
ts = float(a[6]); nref, n, mode = int(a[7]), int(a[8]), int(a[9])
sseed = int(a[10]) if len(a) > 10 else -1
interp = int(a[11]) if len(a) > 11 else 0
norm = int(a[12]) if len(a) > 12 else -1
refs = synth.make_references(nref, nx, ou)
parts, _ = synth.make_particles(refs, n, xr, yr, 0.25, ou=ou)
rg = orc.rings(ir, ou, rs)
mask = orc.model_circle(ou, nx, nx)
refs_n, cref = orc.prepare_refs(refs, mask, rg, interp=interp)
d = np.zeros((n, 2), np.float32)
if sseed >= 0:
    rng = np.random.default_rng(sseed)
    lim = int((nx // 2 + 1 - ou - 2 + 1) / ts)
    d = (rng.integers(-lim, lim + 1, size=(n, 2)) * ts).astype(np.float32)
    for i in range(n):
        parts[i] = np.roll(parts[i], (int(np.floor(d[i, 1])), int(np.floor(d[i, 0]))), axis=(0, 1))
d0 = d.copy()
o_norm = (mode == 0) if norm < 0 else bool(norm)
if mode == 0:
    params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, yr, ts, d, nthreads=8, interp=interp, normalize=o_norm)
else:
    params = np.zeros((n, 6), np.float32)
    params[:, 1:3] = -d0
    params, infos, _, _ = orc.reffree_iteration(parts, cref[0], rg, xr, yr, ts, (0, 0), d, params, nthreads=8, interp=interp)
eng = api.Engine(nx, ou, xr, yr, ts, nref, mode, first_ring=ir, ring_skip=rs, interp=interp, normalize_ring=None if norm < 0 else bool(norm))
eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
st, res = torch.from_numpy(d0.copy()).to(eng.dev), eng.new_result(n)
eng.align(torch.from_numpy(parts).to(eng.dev), st, res)
eng.sync()
r = api.Engine.result_to_numpy(res)
s = st.cpu().numpy()
print("path %d (%d offsets per pass, tiled %d)" % (eng.search_path, eng.search_offsets_per_pass, eng.search_tiled))
bad = 0
for i in range(n):
    same = (r["ref_id"][i], r["mirror"][i], r["angle_bin"][i]) == (int(params[i, 4]), int(params[i, 3]), infos[i].jtot) and abs(s[i] - d[i]).max() < 1e-6
    rel = abs(r["peak"][i] - params[i, 5]) / abs(params[i, 5])
    flag = "" if same and rel < 1e-4 else "   <--"
    bad += bool(flag)
    print("%3d start (%5.1f %5.1f) engine (%d %d %3d) (%5.1f %5.1f) %.6f | oracle (%d %d %3d) (%5.1f %5.1f) %.6f  rel %.2e%s" % (
        i, d0[i, 0], d0[i, 1], r["ref_id"][i], r["mirror"][i], r["angle_bin"][i], s[i, 0], s[i, 1], r["peak"][i],
        int(params[i, 4]), int(params[i, 3]), infos[i].jtot, d[i, 0], d[i, 1], params[i, 5], rel, flag))
print("%d of %d particles differ" % (bad, n))
sys.exit(1 if bad else 0)
