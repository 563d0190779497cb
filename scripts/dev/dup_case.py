import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cryo_ralib_amd import api, synth
from oracle import oracle as orc
nx, ou, xr, nref = (int(v) for v in sys.argv[1:5])
groups = eval(sys.argv[5])
refine = float(sys.argv[6]) if len(sys.argv) > 6 else None
n = 48
refs = synth.make_references(nref, nx, ou)
for g in groups:
    for k in g[1:]:
        refs[k] = refs[g[0]]
parts, truth = synth.make_particles(refs, n, xr, xr, 0.5, ou=ou)
rg = orc.rings(1, ou, 1); mask = orc.model_circle(ou, nx, nx)
refs_n, cref = orc.prepare_refs(refs, mask, rg)
d = np.zeros((n, 2), np.float32)
params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d, nthreads=8)
eng = api.Engine(nx, ou, xr, xr, 1.0, nref, api.RA_MODE_MREF)
if refine is not None: eng.set_refine(refine)
eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
st, res = eng.new_state(n), eng.new_result(n)
eng.align(torch.from_numpy(parts).to(eng.dev), st, res); eng.sync()
r = api.Engine.result_to_numpy(res); s = st.cpu().numpy()
print("path", eng.search_path, "tiled", eng.search_tiled, "refined", eng.last_refine_count())
for i in range(n):
    same = r["ref_id"][i] == int(params[i, 4]) and r["mirror"][i] == int(params[i, 3]) and r["angle_bin"][i] == infos[i].jtot and (s[i] == d[i]).all()
    if not same:
        print("particle %2d class %d: engine ref %d m %d bin %d (%g %g) peak %.6f | oracle ref %d m %d bin %d (%g %g) peak %.6f" % (
            i, truth["cls"][i], r["ref_id"][i], r["mirror"][i], r["angle_bin"][i], s[i, 0], s[i, 1], r["peak"][i],
            int(params[i, 4]), int(params[i, 3]), infos[i].jtot, d[i, 0], d[i, 1], params[i, 5]))
