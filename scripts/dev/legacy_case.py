"""one geometry through the drop-in symbols AND the engine API against the checker: python scripts/dev/legacy_case.py nx ou xr ts nref n"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cryo_ralib_amd import api, synth
from oracle import oracle as orc
nx, ou, xr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); ts = float(sys.argv[4]); nref, n = int(sys.argv[5]), int(sys.argv[6])
lib = api.load_library()
refs = synth.make_references(nref, nx, ou)
parts, _ = synth.make_particles(refs, n, xr, xr, 0.4, ou=ou)
rg = orc.rings(1, ou, 1); mask = orc.model_circle(ou, nx, nx)
refs_n, cref = orc.prepare_refs(refs, mask, rg)
d = np.zeros((n, 2), np.float32)
params, infos, sums, counts = orc.mref_iteration(parts, cref, rg, xr, xr, ts, d, nthreads=16)
cfg = api.AlignConfig(n, nref, nx, ou, rg.maxrin, ts, float(xr), float(xr))
prm = ctypes.cast(lib.pre_align_init(n, ctypes.byref(cfg), 0), api.aln_param_ptr)
lib.pre_align_fetch(api.get_c_ptr_array(list(parts)), n, b"sbj_batch")
lib.reset_shifts(float(xr), ts)
lib.pre_align_fetch(api.get_c_ptr_array(list(refs_n)), nref, b"ref_batch")
lib.mref_align_run_m(0, n)
leg = [(prm[k].ref_id, int(prm[k].mirror), prm[k].shift_x, prm[k].shift_y, prm[k].angle) for k in range(n)]
lib.gpu_clear()
eng = api.Engine(nx, ou, xr, xr, ts, nref, api.RA_MODE_MREF)
eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
st, res = eng.new_state(n), eng.new_result(n)
eng.align(torch.from_numpy(parts).to(eng.dev), st, res); eng.sync()
r = api.Engine.result_to_numpy(res); s = st.cpu().numpy()
print("engine path %d, %d offsets, maxrin %d" % (eng.search_path, eng.num_shifts, eng.maxrin))
for k in range(n):
    flag = "" if (leg[k][2], leg[k][3]) == (d[k, 0], d[k, 1]) and (s[k] == d[k]).all() else "  <--"
    print("%2d legacy ref %d m %d (%5.1f %5.1f) a %.4f | engine ref %d m %d (%5.1f %5.1f) a %.4f peak %.5f | oracle ref %d m %d (%5.1f %5.1f) a %.4f peak %.5f%s" % (
        k, *leg[k], r["ref_id"][k], r["mirror"][k], s[k, 0], s[k, 1], r["alpha"][k], r["peak"][k],
        int(params[k, 4]), int(params[k, 3]), d[k, 0], d[k, 1], params[k, 0], params[k, 5], flag))
print("refine count of the default engine: %d" % eng.last_refine_count())
eng.set_refine(-1.0)
eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
st2, res2 = eng.new_state(n), eng.new_result(n)
eng.align(torch.from_numpy(parts).to(eng.dev), st2, res2); eng.sync()
r2 = api.Engine.result_to_numpy(res2); s2 = st2.cpu().numpy()
for k in range(n):
    if not (s[k] == d[k]).all():
        print("particle %d with every particle refined: (%5.1f %5.1f) ref %d m %d bin %d peak %.6f alpha %.4f; refine count %d" % (
            k, s2[k, 0], s2[k, 1], r2["ref_id"][k], r2["mirror"][k], r2["angle_bin"][k], r2["peak"][k], r2["alpha"][k], eng.last_refine_count()))
# the checker's peak of every offset for the particles that differ (single-offset searches around cnx + ix, cny + iy)
cn = nx // 2 + 1
nk = int(xr / ts)
for k in range(n):
    if (s[k] == d[k]).all():
        continue
    rows = []
    for iy in range(-nk, nk + 1):
        for ix in range(-nk, nk + 1):
            out, info = orc.multiref_polar_ali_2d(parts[k], cref, [0, 0], [0, 0], ts, rg, cn + ix * ts, cn + iy * ts)
            rows.append((float(out[5]), ix * ts, iy * ts, int(out[4]), int(out[3]), info.jtot))
    rows.sort(reverse=True)
    print("particle %d: the checker's best offsets (peak, ix, iy, ref, mirror, jtot):" % k)
    for rrow in rows[:6]:
        print("   %.6f (%4.1f %4.1f) ref %d m %d jtot %d   rel to best %.2e" % (*rrow, (rows[0][0] - rrow[0]) / rows[0][0]))
# the engine's own f32 peak of single offsets (an engine with a one-offset window started at that offset, refinement off)
e1 = api.Engine(nx, ou, 0, 0, ts, nref, api.RA_MODE_MREF)
e1.set_refine(0.0)
e1.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(e1.dev))
for k in range(n):
    if (s[k] == d[k]).all():
        continue
    for off in ((s[k, 0], s[k, 1]), (d[k, 0], d[k, 1])):
        st1 = torch.tensor([[off[0], off[1]]], dtype=torch.float32, device=e1.dev)
        res1 = e1.new_result(1)
        e1.align(torch.from_numpy(parts[k:k + 1].copy()).to(e1.dev), st1, res1); e1.sync()
        q = api.Engine.result_to_numpy(res1)
        print("particle %d, engine f32 peak at offset (%4.1f %4.1f): %.6f ref %d m %d bin %d (path %d)" % (
            k, off[0], off[1], q["peak"][0], q["ref_id"][0], q["mirror"][0], q["angle_bin"][0], e1.search_path))
