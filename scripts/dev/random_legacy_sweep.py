"""random geometries through the DROP-IN symbols of the reference's library (include/ralign.h: pre_align_init, pre_align_fetch,
reset_shifts, mref_align_run_m, pre_align_run_m, get_num_ref, gpu_clear -- the call protocol of test_mref_gpu_align.py:373-449 and
test_reffree_gpu_align.py:330-470) against the CPU checker.  python scripts/dev/random_legacy_sweep.py [ncase] [seed] [small|big|huge]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cryo_ralib_amd import api, synth          # noqa: E402
from oracle import oracle as orc               # noqa: E402

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
size = sys.argv[3] if len(sys.argv) > 3 else "small"
lib = api.load_library()
for case in range(ncase):
    xr = int(rng.integers(1, 4))
    nx = int(rng.integers(140, 200)) if size == "huge" else int(rng.integers(64, 161)) if size == "big" else int(rng.integers(32, 101))
    oumax = (nx - 1) // 2 - xr - 1
    ou = int(rng.integers(61, min(90, oumax) + 1)) if size == "huge" else int(rng.integers(24, min(78, oumax) + 1)) if size == "big" \
        else int(rng.integers(8, min(40, oumax) + 1))
    ts = float(rng.choice([1.0, 1.0, 0.5]))
    mref = rng.random() < 0.6
    n = int(rng.integers(6, 13)) if size == "huge" else int(rng.integers(12, 49))
    nref = max(1, min(int(rng.integers(1, 9)), n // 6)) if mref else 1
    print("case %2d: nx=%d ou=%d xr=%d ts=%g nref=%d n=%d %s" % (case, nx, ou, xr, ts, nref, n, "mref_align_run_m" if mref else "pre_align_run_m"), flush=True)
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 0.4, ou=ou)
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    cfg = api.AlignConfig(n, nref, nx, ou, rg.maxrin, ts, float(xr), float(xr))
    assert lib.pre_align_size_check(n, ctypes.byref(cfg), 0, 0.9, False) is True
    prm = ctypes.cast(lib.pre_align_init(n, ctypes.byref(cfg), 0), api.aln_param_ptr)
    lib.pre_align_fetch(api.get_c_ptr_array(list(parts)), n, b"sbj_batch")
    lib.reset_shifts(float(xr), ts)
    d = np.zeros((n, 2), np.float32)
    if mref:
        refs_n, cref = orc.prepare_refs(refs, mask, rg)
        for it in range(2):
            lib.pre_align_fetch(api.get_c_ptr_array(list(refs_n)), nref, b"ref_batch")
            hs = lib.mref_align_run_m(0, n)
            params, infos, sums, counts = orc.mref_iteration(parts, cref, rg, xr, xr, ts, d, nthreads=16)
            got = np.ctypeslib.as_array(hs, shape=(2, nref, nx, nx))
            cnt = np.ctypeslib.as_array(lib.get_num_ref(), shape=(nref,))
            np.testing.assert_array_equal(cnt, counts)
            for k in range(n):
                assert prm[k].ref_id == int(params[k, 4]) and prm[k].mirror == bool(params[k, 3]), (it, k)
                assert prm[k].shift_x == d[k, 0] and prm[k].shift_y == d[k, 1], (it, k, prm[k].shift_x, prm[k].shift_y, d[k])
            # class sums: the sub-bin angle of the default engine comes from the f32 peak neighbourhood (refine threshold of
            # ra_set_refine), 1e-4 degrees beside the CPU path's on some particles -- more on the coarse rings of a small ou --,
            # and rot_shift2D's interpolant is discontinuous across pixel cells: every such particle moves a few single pixels of
            # its class sum by O(sigma).  The bar is therefore per pixel and counts them: at most 3 pixels per particle of the
            # stack (or 0.1 % of the pixels) beyond 2e-4 of the scale, none of them beyond the size of one particle's pixel
            scale = max(1.0, float(np.abs(sums).max()))
            one = float(np.abs(parts).max())
            for h in (0, 1):
                df = np.abs(got[h] - sums[:, h])
                nbig = int((df > 2e-4 * scale).sum())
                assert nbig <= max(3 * n, df.size // 1000), (it, h, nbig, df.size, n)
                assert df.max() <= 2.0 * one, (it, h, float(df.max()), one)
    else:
        tavg = parts.mean(0)[None].astype(np.float32)
        _, cref = orc.prepare_refs(tavg, None, rg)
        lib.pre_align_fetch(api.get_c_ptr_array(list(tavg)), 1, b"ref_batch")
        assert lib.pre_align_run_m(0, n) != 0
        params = np.zeros((n, 6), np.float32)
        params, infos, _, _ = orc.reffree_iteration(parts, cref[0], rg, xr, xr, ts, (0, 0), d, params, nthreads=16)
        for k in range(n):
            assert prm[k].mirror == bool(params[k, 3]) and prm[k].shift_x == d[k, 0] and prm[k].shift_y == d[k, 1], k
            da = abs(((prm[k].angle - params[k, 0]) + 180.0) % 360.0 - 180.0)
            assert da < 2e-3, (k, prm[k].angle, params[k, 0])
    lib.gpu_clear()
print("all %d cases agree with the checker" % ncase)
