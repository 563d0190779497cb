"""random geometries through the engine against the CPU checker (run on the GPU box: python scripts/dev/random_sweep.py [n] [seed]);
prints one line per case and the search path the engine took; exits non-zero on the first mismatch.  A third argument "big" draws
boxes of 64 .. 160 pixels with rings up to 512 samples and up to 60 references (search_pair / search_duo / the generic kernels);
"huge" boxes of 140 .. 230 pixels.  A fourth argument "state" starts every particle from a random accumulated shift within
+-(mashi + 1) (multiples of the step): shrunken and reset search windows, the live-offset lists of the size-generic class."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cryo_ralib_amd import api, synth          # noqa: E402
from oracle import oracle as orc               # noqa: E402
from test_gpu_parity import compare_search     # noqa: E402

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
big = len(sys.argv) > 3 and sys.argv[3] in ("big", "huge")
huge = len(sys.argv) > 3 and sys.argv[3] == "huge"      # boxes of 140 .. 230 pixels, ou 61 .. 100: the size-generic class (polar_zone_kernel)
wide = len(sys.argv) > 5 and sys.argv[5] == "wide"      # odd steps, fractional ranges, fractional states, up to 130 references
rand_state = len(sys.argv) > 4 and sys.argv[4] in ("state", "state+options")
rand_opt = len(sys.argv) > 4 and sys.argv[4] in ("options", "state+options")      # Util::quadri sampling / Normalize_ring against the mode
for case in range(ncase):
    xr = int(rng.integers(0, 4)); yr = int(rng.integers(0, 4))
    nx = int(rng.integers(140, 231)) if huge else int(rng.integers(64, 161)) if big else int(rng.integers(36, 101))
    oumax = (nx - 1) // 2 - max(xr, yr) - 2
    ou = int(rng.integers(61, min(100, oumax) + 1)) if huge else int(rng.integers(24, min(78, oumax) + 1)) if big else int(rng.integers(8, min(40, oumax) + 1))
    ir = int(rng.integers(1, 4)); rs = int(rng.integers(1, 3))
    ts = float(rng.choice([1.0, 1.0, 0.5])) if not wide else float(rng.choice([1.0, 0.5, 0.25, 0.75, 1.5, 2.0, 3.0]))
    if wide and rng.random() < 0.4:          # fractional ranges: int(range / step) offsets, the range itself in search_range
        xr = float(xr) + float(rng.choice([0.25, 0.5, 0.8])); yr = float(yr) + float(rng.choice([0.0, 0.5, 0.8]))
    mode = api.RA_MODE_MREF if rng.random() < 0.7 else api.RA_MODE_REFFREE
    nref = (int(rng.integers(1, 21)) if huge else int(rng.integers(1, 61)) if big else int(rng.integers(1, 17))) if mode == api.RA_MODE_MREF else 1
    if wide and mode == api.RA_MODE_MREF and not huge and rng.random() < 0.25:
        nref = int(rng.integers(60, 131))
    n = int(rng.integers(2, 6)) if huge else int(rng.integers(3, 9)) if big else int(rng.integers(3, 20))
    interp, norm = api.RA_INTERP_BILINEAR, -1
    if rand_opt:
        interp = api.RA_INTERP_QUADRI if rng.random() < 0.4 else api.RA_INTERP_BILINEAR
        norm = int(rng.choice([-1, 0, 1])) if mode == api.RA_MODE_MREF else -1
    nomirror = bool(rand_opt and mode == api.RA_MODE_REFFREE and rng.random() < 0.4)      # --nomirror: ormq(nomirror) -> Crosrng_ns
    o_interp = orc.INTERP_QUADRI if interp == api.RA_INTERP_QUADRI else orc.INTERP_BILINEAR
    o_norm = (mode == api.RA_MODE_MREF) if norm < 0 else bool(norm)
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, yr, 0.25, ou=ou)
    rg = orc.rings(ir, ou, rs)
    mask = orc.model_circle(ou, nx, nx)
    refs_n, cref = orc.prepare_refs(refs, mask, rg, interp=o_interp) if rand_opt else orc.prepare_refs(refs, mask, rg)
    d = np.zeros((n, 2), np.float32)
    if rand_state:
        mashi = nx // 2 + 1 - ou - 2
        lim = int((mashi + 1) / ts)
        d = (rng.integers(-lim, lim + 1, size=(n, 2)) * ts).astype(np.float32)
        if wide:          # states off the step grid (the centre correction of ali2d_single_iter leaves arbitrary fractions)
            d = (d + (rng.random((n, 2)) < 0.5) * rng.uniform(-0.5, 0.5, size=(n, 2))).astype(np.float32)
    d0 = d.copy()
    if rand_state:      # the particle sits where its state says (a search centred 20 pixels beside it would only find noise peaks near zero)
        for i in range(n):
            if mode == api.RA_MODE_MREF and np.abs(d0[i]).max() > nx // 2 + 1 - ou - 2:
                continue          # mref_ali2d resets this state: the search runs around the centre, where the particle still is
            parts[i] = np.roll(parts[i], (int(np.floor(d0[i, 1])), int(np.floor(d0[i, 0]))), axis=(0, 1))
    if mode == api.RA_MODE_MREF:
        params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, yr, ts, d, nthreads=8, interp=o_interp, normalize=o_norm)
    else:
        params = np.zeros((n, 6), np.float32)
        params[:, 1:3] = -d0
        orc.set_nomirror(nomirror)
        try:
            params, infos, _, _ = orc.reffree_iteration(parts, cref[0], rg, xr, yr, ts, (0, 0), d, params, nthreads=8, interp=o_interp)
        finally:
            orc.set_nomirror(False)
    eng = api.Engine(nx, ou, xr, yr, ts, nref, mode, first_ring=ir, ring_skip=rs, interp=interp, normalize_ring=None if norm < 0 else bool(norm))
    eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
    if nomirror:
        eng.set_nomirror(True)
    tp = torch.from_numpy(parts).to(eng.dev)
    st, res = torch.from_numpy(d0.copy()).to(eng.dev), eng.new_result(n)
    eng.align(tp, st, res)
    eng.sync()
    path = "%d (%d offsets per pass)" % (eng.search_path, eng.search_offsets_per_pass)
    try:
        compare_search(api.Engine.result_to_numpy(res), st.cpu().numpy(), params, infos, d)
    except AssertionError:
        r = api.Engine.result_to_numpy(res)
        print("case %d FAILED: nx=%d ou=%d ir=%d rs=%d xr=%g yr=%g ts=%g nref=%d n=%d mode=%d path=%s" % (case, nx, ou, ir, rs, xr, yr, ts, nref, n, mode, path))
        print(" start states", d0.tolist())
        print(" peaks engine", r["peak"].tolist())
        print(" peaks oracle", params[:, 5].tolist())
        print(" assignment engine", list(zip(r["ref_id"].tolist(), r["mirror"].tolist(), r["angle_bin"].tolist())), st.cpu().numpy().tolist())
        print(" assignment oracle", [(int(params[i, 4]), int(params[i, 3]), infos[i].jtot) for i in range(n)], d.tolist())
        raise
    eng.close()
    print("case %2d ok: nx=%d ou=%d ir=%d rs=%d xr=%g yr=%g ts=%g nref=%d n=%d mode=%d interp=%d norm=%d nomirror=%d path=%s" % (case, nx, ou, ir, rs, xr, yr, ts, nref, n, mode, interp, norm, nomirror, path), flush=True)
print("all %d cases agree with the checker" % ncase)
