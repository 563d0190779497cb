"""random geometries through the HOST DRIVERS (MrefAligner / RefFreeAligner: search, class sums, reference update, state round trip)
for three iterations, every iteration against the same step built from oracle calls on the driver's own inputs of that iteration
(its current references, previous parameters and state), so a float tie in one iteration cannot drift into the next.
Run on the GPU box: python scripts/dev/random_loop_sweep.py [ncase] [seed] [small|big|huge].  Exits non-zero on the first mismatch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cryo_ralib_amd import api, synth                               # noqa: E402
from cryo_ralib_amd.mref import MrefAligner, RefFreeAligner         # noqa: E402
from oracle import oracle as orc                                    # noqa: E402
from test_gpu_parity import compare_search, assert_images_close     # noqa: E402

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
size = sys.argv[3] if len(sys.argv) > 3 else "small"
NIT = 3


def alpha_to_the_ulp(got, want):
    """tests/test_gpu_parity.py::assert_alpha_equal_to_the_ulp with a count floor for the small stacks of this sweep: at most two ulp
    (6.2e-5 degrees near 360), on at most max(2, 3 %) of the particles (device sin / cos in double against libm's)"""
    da = np.abs(((got.astype(np.float64) - want.astype(np.float64)) + 180.0) % 360.0 - 180.0)
    assert da.max() <= 6.2e-5, da.max()
    assert (got != want).sum() <= max(2, int(0.03 * len(got))), (got != want).sum()


def params6(r):
    p = np.zeros((len(r), 6), np.float32)
    p[:, 0] = r["alpha"]; p[:, 1] = r["sx"]; p[:, 2] = r["sy"]; p[:, 3] = r["mirror"]
    return p


for case in range(ncase):
    xr = int(rng.integers(0, 4)); yr = int(rng.integers(0, 4))
    if size == "huge":
        nx = int(rng.integers(140, 200))
    elif size == "big":
        nx = int(rng.integers(64, 161))
    else:
        nx = int(rng.integers(36, 101))
    oumax = (nx - 1) // 2 - max(xr, yr) - 1
    if size == "huge":
        ou = int(rng.integers(61, min(90, oumax) + 1))
    elif size == "big":
        ou = int(rng.integers(24, min(78, oumax) + 1))
    else:
        ou = int(rng.integers(8, min(40, oumax) + 1))
    ir = int(rng.integers(1, 4)); rs = int(rng.integers(1, 4))
    ts = float(rng.choice([1.0, 1.0, 0.5, 2.0]))
    mref = rng.random() < 0.6
    n = int(rng.integers(8, 17)) if size == "huge" else int(rng.integers(24, 73))
    nref = max(1, min(int(rng.integers(1, 13)), n // 8)) if mref else 1
    chunk = int(rng.choice([0, 0, 16, 24]))
    roundtrip = bool(rng.random() < 0.6)
    tag = "nx=%d ou=%d ir=%d rs=%d xr=%d yr=%d ts=%g nref=%d n=%d chunk=%d %s" % (
        nx, ou, ir, rs, xr, yr, ts, nref, n, chunk, ("mref" + (" round trip" if roundtrip else " exact carry")) if mref else "reference-free")
    print("case %2d: %s" % (case, tag), flush=True)
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, max(xr, 1), max(yr, 1), 0.4, ou=ou)
    rg = orc.rings(ir, ou, rs)
    mask = orc.model_circle(ou, nx, nx)
    if mref:
        al = MrefAligner(parts, refs, ou, xr, yr, ts, ir=ir, rs=rs, preprocess=True, state_roundtrip=roundtrip, refine=-1, chunk=chunk)
        op = np.stack([orc.normalize_mask(p, mask, 0) for p in parts])
        np.testing.assert_allclose(al.particles.cpu().numpy(), op, atol=3e-6 * max(1.0, np.abs(op).max()))
        op = al.particles.cpu().numpy()
        for it in range(NIT):
            cur = al.refs.cpu().numpy().copy()
            d_before = al.state.cpu().numpy().copy()
            prev = params6(al.params()) if it > 0 else None
            _, cref = orc.prepare_refs(cur, None, rg)
            d = orc.state_from_params(prev, 0) if (roundtrip and prev is not None) else d_before
            params, infos, sums, counts = orc.mref_iteration(op, cref, rg, xr, yr, ts, d, nthreads=16)
            got_counts = al.iterate()
            r = al.params()
            assert compare_search(r, al.state.cpu().numpy(), params, infos, d) == 0
            np.testing.assert_array_equal(got_counts, counts)
            alpha_to_the_ulp(r["alpha"], params[:, 0])
            if counts.min() < 4:
                print("   iteration %d: a class vanished (reseeded from the driver's RNG), case ends" % it)
                break
            want = np.stack([orc.normalize_mask((sums[j, 0] + sums[j, 1]) * np.float32(1.0 / float(counts[j])), mask, 1) for j in range(nref)])
            got = al.refs.cpu().numpy()
            # (rot_shift2D's interpolant is discontinuous across pixel cells: a particle whose alpha is one ulp beside the checker's
            # moves single pixels of its class average by O(sigma / class size); with every alpha equal the averages agree to 1e-5)
            nmis = int(((r["alpha"] != params[:, 0]) | (r["sx"] != params[:, 1]) | (r["sy"] != params[:, 2])).sum())
            df = np.abs(got - want)
            sc = float(np.abs(want).max())
            if nmis == 0:
                if df.max() > 1e-5 * sc:
                    bad = np.argwhere(df > 1e-5 * sc)
                    print("   iteration %d: every parameter equal, yet %d pixels of the averages differ (max %.4g of %.4g):" % (it, len(bad), df.max(), sc), bad[:8].tolist())
                    print("   class sizes", counts.tolist())
                assert df.max() <= 1e-5 * sc, (float(df.max()), sc)
            else:
                assert int((df > 1e-4 * sc).sum()) <= 6 * nmis, (int((df > 1e-4 * sc).sum()), nmis)
                assert np.quantile(df, 0.99) <= 1e-4 * sc
        print("   path %d (%d offsets per pass), %d iterations ok" % (al.engine.search_path, al.engine.search_offsets_per_pass, it + 1), flush=True)
        al.close()
    else:
        # a second stage on some cases (--xr "3 2" --ts "1 0.5": reset_shifts between the stages, test_reffree_gpu_align.py:355-357)
        two = rng.random() < 0.4
        xr2, yr2, ts2 = max(xr - 1, 0), max(yr - 1, 0), ts / 2
        al = RefFreeAligner(parts, ou, "%d %d" % (xr, xr2) if two else str(xr), "%d %d" % (yr, yr2) if two else str(yr),
                            "%g %g" % (ts, ts2) if two else str(ts), ir=ir, rs=rs, chunk=chunk, refine=-1)
        center = -1 if rng.random() < 0.5 else 0          # the average-centre rule: states off the step grid from the second iteration on
        xr1, yr1, ts1 = xr, yr, ts
        for it in range(NIT):
            if two and it >= 1:
                al.set_stage(1)
                xr, yr, ts = xr2, yr2, ts2
            prev = params6(al.params())
            d = al.state.cpu().numpy().copy()
            al.iterate(center, None)
            al.engine.sync()
            _, cref = orc.prepare_refs(al.tavg.cpu().numpy(), None, rg)
            params, infos, sums, _ = orc.reffree_iteration(parts, cref[0], rg, xr, yr, ts, al.cs, d, prev.copy(), nthreads=16)
            r = al.params()
            assert compare_search(r, al.state.cpu().numpy(), params, infos, d) == 0
            alpha_to_the_ulp(r["alpha"], params[:, 0])
            got = (al.buf.sums[0, 0] + al.buf.sums[0, 1]).cpu().numpy()
            want = sums[0, 0] + sums[0, 1]
            # under the mask: the bar of the tests; outside it rot_shift2D copies the input pixel wherever the source position leaves
            # the image ("background"), a jump along the rotated frame that a one-ulp sx / sy moves across single pixels
            sc = float(np.abs(want).max())
            assert_images_close(got, want, mask, 1e-4 * sc)
            df = np.abs(got - want)
            nbig = int((df > 1e-4 * sc).sum())
            assert nbig <= max(4, df.size // 200), (nbig, df.size, float(df.max()), sc)
        print("   path %d (%d offsets per pass), %d iterations ok (center %d, last cs %.3f %.3f%s)" % (al.engine.search_path, al.engine.search_offsets_per_pass, NIT, center, al.cs[0], al.cs[1], ", second stage xr %d yr %d ts %g" % (xr2, yr2, ts2) if two else ""), flush=True)
        xr, yr, ts = xr1, yr1, ts1
        al.close()
print("all %d cases agree with the checker" % ncase)
