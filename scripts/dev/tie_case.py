"""one multi-reference geometry, several iterations through the engine API against the checker; for every particle that differs: the
checker's peak of every search offset.  python scripts/dev/tie_case.py nx ou xr ts nref n [iterations] [sigma]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cryo_ralib_amd import api, synth
from oracle import oracle as orc
nx, ou, xr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); ts = float(sys.argv[4]); nref, n = int(sys.argv[5]), int(sys.argv[6])
nit = int(sys.argv[7]) if len(sys.argv) > 7 else 1
sigma = float(sys.argv[8]) if len(sys.argv) > 8 else 0.4
refs = synth.make_references(nref, nx, ou)
parts, _ = synth.make_particles(refs, n, xr, xr, sigma, ou=ou)
rg = orc.rings(1, ou, 1); mask = orc.model_circle(ou, nx, nx)
refs_n, cref = orc.prepare_refs(refs, mask, rg)
eng = api.Engine(nx, ou, xr, xr, ts, nref, api.RA_MODE_MREF)
eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
tp = torch.from_numpy(parts).to(eng.dev)
st, res = eng.new_state(n), eng.new_result(n)
d = np.zeros((n, 2), np.float32)
cn = nx // 2 + 1
nk = int(xr / ts)
bad = 0
lib = api.load_library()
cfg = api.AlignConfig(n, nref, nx, ou, rg.maxrin, ts, float(xr), float(xr))
prm = ctypes.cast(lib.pre_align_init(n, ctypes.byref(cfg), 0), api.aln_param_ptr)
lib.pre_align_fetch(api.get_c_ptr_array(list(parts)), n, b"sbj_batch")
lib.reset_shifts(float(xr), ts)
for it in range(nit):
    d_in = d.copy()
    params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, ts, d, nthreads=16)
    eng.align(tp, st, res); eng.sync()
    r = api.Engine.result_to_numpy(res); s = st.cpu().numpy()
    print("iteration %d: refine count %d" % (it, eng.last_refine_count()))
    lib.pre_align_fetch(api.get_c_ptr_array(list(refs_n)), nref, b"ref_batch")
    lib.mref_align_run_m(0, n)
    for k in range(n):
        if not (prm[k].shift_x == d[k, 0] and prm[k].shift_y == d[k, 1] and prm[k].ref_id == int(params[k, 4]) and prm[k].mirror == bool(params[k, 3])):
            print("  drop-in symbols, particle %d from (%4.1f %4.1f): (%4.1f %4.1f) ref %d m %d angle %.4f | checker (%4.1f %4.1f) ref %d m %d alpha %.4f peak %.6f" % (
                k, d_in[k, 0], d_in[k, 1], prm[k].shift_x, prm[k].shift_y, prm[k].ref_id, prm[k].mirror, prm[k].angle,
                d[k, 0], d[k, 1], int(params[k, 4]), int(params[k, 3]), params[k, 0], params[k, 5]))
    for k in range(n):
        if (s[k] == d[k]).all() and r["ref_id"][k] == int(params[k, 4]) and r["mirror"][k] == int(params[k, 3]) and r["angle_bin"][k] == infos[k].jtot:
            continue
        bad += 1
        print("  particle %d from (%4.1f %4.1f): engine (%4.1f %4.1f) ref %d m %d bin %d peak %.6f | checker (%4.1f %4.1f) ref %d m %d bin %d peak %.6f" % (
            k, d_in[k, 0], d_in[k, 1], s[k, 0], s[k, 1], r["ref_id"][k], r["mirror"][k], r["angle_bin"][k], r["peak"][k],
            d[k, 0], d[k, 1], int(params[k, 4]), int(params[k, 3]), infos[k].jtot, params[k, 5]))
        rows = []
        for iy in range(-nk, nk + 1):
            for ix in range(-nk, nk + 1):
                out, info = orc.multiref_polar_ali_2d(parts[k], cref, [0, 0], [0, 0], ts, rg, np.float32(cn + d_in[k, 0]) + np.float32(ix * ts), np.float32(cn + d_in[k, 1]) + np.float32(iy * ts))
                rows.append((float(out[5]), d_in[k, 0] + ix * ts, d_in[k, 1] + iy * ts, int(out[4]), int(out[3]), info.jtot))
        rows.sort(reverse=True)
        for rrow in rows[:5]:
            print("     checker %.6f (%4.1f %4.1f) ref %d m %d jtot %d   rel to best %.2e" % (*rrow, (rows[0][0] - rrow[0]) / rows[0][0]))
    st.copy_(torch.from_numpy(d))          # continue from the checker's state
print("%d disagreements" % bad)
