"""degenerate particles in a batch -- all zeros, a constant, NaN, inf, 1e30 -- through every kernel family: the call must return, and the
ordinary particles beside them must get the results they get without them.  python scripts/dev/degenerate_inputs.py"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cryo_ralib_amd import api, synth

GEOS = [(90, 36, 3, 10, 0), (90, 36, 3, 1, 1), (90, 36, 3, 50, 0), (44, 10, 2, 3, 0), (100, 40, 3, 10, 0), (130, 52, 2, 6, 0),
        (128, 60, 2, 4, 1), (150, 66, 2, 3, 0), (256, 36, 3, 10, 0)]
for nx, ou, xr, nref, mode in GEOS:
    n = 24
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 0.5, ou=ou)
    bad = parts.copy()
    bad[3] = 0.0
    bad[7] = 2.5
    bad[11] = np.nan
    bad[12, nx // 2, nx // 2] = np.inf
    bad[17] *= 1e30
    bad[20, nx // 2 - 3, nx // 2 + 2] = np.nan
    special = [3, 7, 11, 12, 17, 20]
    out = []
    for stack in (parts, bad):
        eng = api.Engine(nx, ou, xr, xr, 1.0, nref, mode)
        eng.set_references(torch.from_numpy(refs).to(eng.dev))
        st, res = eng.new_state(n), eng.new_result(n)
        eng.align(torch.from_numpy(stack).to(eng.dev), st, res)
        eng.sync()
        out.append((api.Engine.result_to_numpy(res).copy(), st.cpu().numpy().copy()))
        path = eng.search_path
        # class sums of the same call must not fault either
        eng.close()
    keep = [i for i in range(n) if i not in special]
    for f in api.RESULT_DTYPE.names:
        np.testing.assert_array_equal(out[0][0][f][keep], out[1][0][f][keep], err_msg="%s at %d/%d" % (f, nx, ou))
    np.testing.assert_array_equal(out[0][1][keep], out[1][1][keep])
    r = out[1][0]
    print("nx=%d ou=%d nref=%d mode=%d path %d: ordinary particles unchanged; degenerate ones ->" % (nx, ou, nref, mode, path),
          [(int(i), int(r["ref_id"][i]), int(r["mirror"][i]), int(r["angle_bin"][i]), float(r["peak"][i])) for i in special], flush=True)
print("done")
