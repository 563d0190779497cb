"""CPU tests of the oracle (oracle/ralign_oracle.c): pinned against the known answers the
reference tree holds, against independent numpy formulations, and against planted truth."""
import json
import os

import numpy as np
import pytest

from cryo_ralib_amd import geometry, synth
from oracle import oracle as orc


@pytest.fixture(scope="module")
def known(golden_dir):
    with open(os.path.join(golden_dir, "known_answers.json")) as f:
        return json.load(f)


def test_combine_params2_known_answer(known):
    for case in known["combine_params2"]:
        out = orc.combine_params2(*case["args"])
        np.testing.assert_allclose(out[:3], case["out"][:3], rtol=0, atol=1e-6)
        assert out[3] == case["out"][3]


def test_inverse_transform2_known_answer(known):
    for case in known["inverse_transform2"]:
        out = orc.inverse_transform2(*case["args"])
        # the notebook value went through EMAN2's float Transform matrix
        np.testing.assert_allclose(out[:3], case["out"], rtol=0, atol=1e-5)


def test_transform_algebra_with_mirror():
    # inverse(T) o T = identity and associativity with mirrors, v' = M (R v + t)
    rng = np.random.default_rng(0)
    for _ in range(50):
        a, sx, sy = rng.uniform(0, 360), rng.uniform(-5, 5), rng.uniform(-5, 5)
        m = int(rng.integers(0, 2))
        ia, isx, isy, im = orc.inverse_transform2(a, sx, sy, m)
        ca, csx, csy, cm = orc.combine_params2(a, sx, sy, m, ia, isx, isy, im)
        assert cm == 0
        assert min(abs(ca), abs(ca - 360)) < 1e-9 and abs(csx) < 1e-9 and abs(csy) < 1e-9
        g = geometry.combine_params2(a, sx, sy, m, ia, isx, isy, im)
        assert abs(g[1]) < 1e-9 and abs(g[2]) < 1e-9 and g[3] == 0


def test_prb1d_coefficients(known):
    c2 = np.array(known["prb1d_coefficients"]["c2"], float)
    c3 = np.array(known["prb1d_coefficients"]["c3"], float)
    rng = np.random.default_rng(1)
    for _ in range(20):
        b = rng.normal(size=7)
        expect = np.float32(c2 @ b / (2 * (c3 @ b)) - 4)
        assert orc.prb1d7(b) == pytest.approx(expect, rel=1e-6)
    # exact parabola: vertex recovered
    x = np.arange(-3, 4, dtype=float)
    assert orc.prb1d7(-(x - 0.3) ** 2) == pytest.approx(0.3, abs=1e-6)
    assert orc.prb1d7(np.zeros(7)) == 0.0


def test_numrinit_ringwe(known):
    rg = orc.rings(1, 36, 1)
    numr = rg.numr_list()
    lens = numr[2::3]
    want = {int(k): v for k, v in known["numrinit_ou36"]["lengths"].items()}
    assert {n: lens.count(n) for n in set(lens)} == want
    assert rg.lcirc == known["numrinit_ou36"]["lcirc"] and rg.maxrin == known["numrinit_ou36"]["maxrin"]
    assert numr == geometry.numrinit(1, 36, 1)
    np.testing.assert_allclose(rg.wr_list(), geometry.ringwe(numr), rtol=1e-6)
    for first, last, skip in [(1, 12, 1), (2, 30, 2), (1, 120, 1), (3, 52, 1)]:
        assert orc.rings(first, last, skip).numr_list() == geometry.numrinit(first, last, skip)


def test_ang_n():
    assert orc.ang_n(1.0, 256) == 0.0
    assert orc.ang_n(65.0, 256) == pytest.approx(90.0)
    assert orc.ang_n(256.5, 256) == pytest.approx(359.296875)
    assert 0.0 <= orc.ang_n(0.7, 256) < 360.0


def test_frngs_matches_numpy_rfft():
    rng = np.random.default_rng(2)
    rg = orc.rings(1, 36, 1)
    circ = rng.normal(size=rg.lcirc).astype(np.float32)
    F = orc.frngs(circ, rg)
    numr = rg.numr_list()
    for i in range(rg.nring):
        o, n = numr[3 * i + 1] - 1, numr[3 * i + 2]
        X = np.fft.rfft(circ[o:o + n].astype(np.float64))
        pk = np.zeros(n)
        pk[0], pk[1] = X[0].real, X[n // 2].real
        pk[2::2], pk[3::2] = X[1:n // 2].real, X[1:n // 2].imag
        np.testing.assert_allclose(F[o:o + n], pk, atol=2e-5 * np.abs(pk).max())


def test_crosrng_ms_is_circular_correlation():
    """q[j] = sum_rings w_r sum_i ref(i+j) img(i) (trigonometric interpolation of shorter rings);
    checked on equal-length rings where it is an exact circular correlation."""
    rng = np.random.default_rng(3)
    rg = orc.rings(30, 36, 1)          # all rings length 256
    numr = rg.numr_list()
    assert set(numr[2::3]) == {256}
    a = rng.normal(size=rg.lcirc).astype(np.float32)
    b = rng.normal(size=rg.lcirc).astype(np.float32)
    ca = orc.applyws(orc.frngs(a, rg), rg)
    cb = orc.frngs(b, rg)
    r = orc.crosrng_ms(ca, cb, rg)
    q = np.zeros(256); t = np.zeros(256)
    wr = rg.wr_list()
    for i in range(rg.nring):
        o = numr[3 * i + 1] - 1
        x, y = a[o:o + 256].astype(float), b[o:o + 256].astype(float)
        for j in range(256):
            q[j] += wr[i] * np.dot(np.roll(x, -j), y)
            t[j] += wr[i] * np.dot(np.roll(x[::-1], 1 - j), y)   # mirrored: sum_i x(-i-j) y(i)
    assert r["jn"] - 1 == int(np.argmax(q))
    assert r["qn"] == pytest.approx(q.max(), rel=2e-5)
    assert r["jm"] - 1 == int(np.argmax(t))
    assert r["qm"] == pytest.approx(t.max(), rel=2e-5)


def test_crosrng_last_maximum_wins():
    # a constant CCF: every bin ties, EMAN2's ">=" scan keeps the last one (jtot = maxrin)
    rg = orc.rings(30, 36, 1)
    z = np.zeros(rg.lcirc, np.float32)
    r = orc.crosrng_ms(z, z, rg)
    assert r["jn"] == 256 and r["jm"] == 256 and r["qn"] == 0.0


def test_polar2dm_bilinear_on_linear_image():
    # bilinear interpolation reproduces a plane exactly; sample j sits at (cnx + r sin, cny + r cos)
    nx = 64
    yy, xx = np.mgrid[0:nx, 0:nx].astype(np.float32)
    img = (0.5 * xx - 0.25 * yy + 3).astype(np.float32)
    rg = orc.rings(1, 20, 1)
    cnx = cny = nx // 2 + 1
    circ = orc.polar2dm(img, cnx + 1.0, cny - 2.0, rg)
    numr = rg.numr_list()
    for i in range(rg.nring):
        r, o, n = numr[3 * i], numr[3 * i + 1] - 1, numr[3 * i + 2]
        phi = 2 * np.pi * np.arange(n) / n
        x0 = (cnx + 1.0 - 1) + r * np.sin(phi)       # 0-based
        y0 = (cny - 2.0 - 1) + r * np.cos(phi)
        np.testing.assert_allclose(circ[o:o + n], 0.5 * x0 - 0.25 * y0 + 3, atol=2e-4)


def test_normalize_ring_weighted_stats():
    rng = np.random.default_rng(4)
    rg = orc.rings(1, 36, 1)
    circ = (rng.normal(size=rg.lcirc) * 3 + 7).astype(np.float32)
    out = orc.normalize_ring(circ, rg)
    numr = rg.numr_list()
    w = np.concatenate([np.full(numr[3 * i + 2], numr[3 * i] * 2 * np.pi / numr[3 * i + 2]) for i in range(rg.nring)])
    mean = np.sum(w * out) / w.sum()
    var = np.sum(w * out * out) / w.sum() - mean ** 2
    assert abs(mean) < 1e-4 and var == pytest.approx(1.0, abs=1e-3)


def test_model_circle_and_normalize_mask():
    m = orc.model_circle(36, 90, 90)
    assert (m == geometry.model_circle(36, 90, 90)).all()
    assert m[45, 45 + 36] == 1 and m[45, 45 + 37] == 0 and m[45 - 36, 45] == 1
    rng = np.random.default_rng(5)
    img = rng.normal(2, 3, (90, 90)).astype(np.float32)
    a = orc.normalize_mask(img, m, 0)
    assert abs(a[m > 0.5].mean()) < 1e-5 and a.std() == pytest.approx(img.std(), rel=1e-5)
    b = orc.normalize_mask(img, m, 1)
    assert abs(b[m > 0.5].mean()) < 1e-5 and b[m > 0.5].std(ddof=1) == pytest.approx(1.0, abs=1e-5)
    np.testing.assert_allclose(b, geometry.normalize_mask(img, m, 1), atol=1e-5)


def test_rot_shift2d_identity_and_mirror_rule():
    rng = np.random.default_rng(6)
    for nx in (32, 33):
        img = rng.normal(size=(nx, nx)).astype(np.float32)
        np.testing.assert_array_equal(orc.rot_shift2d(img, 0, 0, 0, 0), img)
        mir = orc.rot_shift2d(img, 0, 0, 0, 1)
        start = 1 - nx % 2                       # notebook 02 cell 2: out[:, start:] = flip(out[:, start:])
        want = img.copy(); want[:, start:] = want[:, start:][:, ::-1]
        np.testing.assert_array_equal(mir, want)
        # integer shift moves content by (+sx, +sy)
        sh = orc.rot_shift2d(img, 0, 2, -1, 0)
        np.testing.assert_allclose(sh[5:-5, 5:-5], np.roll(np.roll(img, 2, 1), -1, 0)[5:-5, 5:-5], atol=1e-6)


def test_rot_shift2d_matches_the_reference_tree_bit_for_bit(golden_dir):
    """tests/golden/rot_shift2d_ref.npz holds what the reference tree's own text of rot_scale_trans2D_background /
    quadri_background (notebook/02 cell 2) computes, compiled as host C++ by tests/golden/make_reftree_pins.py.
    The oracle must reproduce it exactly."""
    g = np.load(os.path.join(golden_dir, "rot_shift2d_ref.npz"))
    assert int(g["ncase"]) == 3
    for k in range(int(g["ncase"])):
        img, ang, dx, dy, out = (g["%s%d" % (n, k)] for n in ("img", "ang", "dx", "dy", "out"))
        for i in range(len(img)):
            got = orc.rot_shift2d(img[i], float(ang[i]), float(dx[i]), float(dy[i]), 0)
            np.testing.assert_array_equal(got, out[i])
            # mirror = the notebook's python line applied to the same output
            start = 1 - img.shape[-1] % 2
            want = out[i].copy(); want[:, start:] = want[:, start:][:, ::-1]
            np.testing.assert_array_equal(orc.rot_shift2d(img[i], float(ang[i]), float(dx[i]), float(dy[i]), 1), want)
            if abs(dx[i]) < img.shape[-1]:      # the generator's numpy twin (synthetic stacks only) agrees to rounding
                np.testing.assert_allclose(synth.rot_shift2d_np(img[i], ang[i], dx[i], dy[i], 0), out[i], atol=2e-6)


def test_search_range_rule():
    # n=90, cn=46, radius 36: ql = 8 + s, qe = 8 - s, returned [left, right] after the caller's swap
    assert orc.search_range(90, 36, 0.0, 3.0) == [3.0, 3.0]
    assert orc.search_range(90, 36, 7.0, 3.0) == [3.0, 1.0]
    assert orc.search_range(90, 36, -6.0, 3.0) == [2.0, 3.0]
    assert orc.search_range(90, 36, 9.0, 3.0) == [3.0, 0.0]
    assert geometry.search_range(90, 36, 7.0, 3.0) == [3.0, 1.0]


def test_window_radius_with_a_ring_step():
    """rings 2, 4 .. 14 of last_ring = 15 (Numrinit(2, 15, 2)): mref_ali2d takes mashi and search_range from its last_ring ARGUMENT
    (test_mref_gpu_align.py:740, 761-766: mashi = 37 - 15 - 2 = 20), ali2d_single_iter from ou = numr[-3] = 14 (mashi = 21,
    clamped not reset) -- a state of 20.5 is reset by the first and kept by the second"""
    nx, ou, n = 73, 15, 3
    refs = synth.make_references(2, nx, ou)
    parts, _ = synth.make_particles(refs, n, 1, 1, 0.1, ou=ou)
    rg = orc.rings(2, ou, 2)
    assert rg.numr[3 * (rg.nring - 1)] == 14 and rg.last_ring == 15
    mask = orc.model_circle(ou, nx, nx)
    _, cref = orc.prepare_refs(refs, mask, rg)
    # mref: 20.5 > 20 is reset -> the result of a particle that starts from (0, 0); 20 is not
    d_edge = np.array([[20.5, 0], [0, -20.5], [20, 0]], np.float32)
    d_zero = np.array([[0, 0], [0, 0], [20, 0]], np.float32)
    pe, _, _, _ = orc.mref_iteration(parts, cref, rg, 1, 1, 1.0, d_edge)
    pz, _, _, _ = orc.mref_iteration(parts, cref, rg, 1, 1, 1.0, d_zero)
    np.testing.assert_array_equal(pe, pz)
    np.testing.assert_array_equal(d_edge, d_zero)
    assert abs(d_edge[2, 0]) >= 19
    # reference-free: 20.5 <= 21 is kept (the search runs around it), 21.5 is clamped to 21
    def run(d0):
        d = np.array(d0, np.float32)
        p = np.zeros((n, 6), np.float32)
        p[:, 1:3] = -d
        out = orc.reffree_iteration(parts, cref[0], rg, 1, 1, 1.0, (0, 0), d, p)
        return out[0], d
    p_in, d_in = run([[20.5, 0], [0, 21.5], [0, 0]])
    p_cl, d_cl = run([[20.5, 0], [0, 21.0], [0, 0]])
    assert abs(d_in[0, 0] - 20.5) <= 1.0 and abs(d_in[1, 1] - 21.0) <= 1.0
    np.testing.assert_array_equal(d_in, d_cl)
    np.testing.assert_array_equal(p_in[:, [0, 3, 5]], p_cl[:, [0, 3, 5]])


def test_float_running_peak_of_the_multireference_scan():
    """Util::multiref_polar_ali_2d keeps its running `peak` as a float (peak = static_cast<float>(qn)) and compares the next
    candidate's double with the rounded value.  With three copies of a reference in the stack (equal CCFs to the bit) the scan
    therefore keeps the FIRST copy when its peak was rounded up and ends on the LAST one otherwise -- never on the middle one;
    both outcomes occur.  ormq with Normalize_ring (orc_set_ormq_normalize) finds the same peaks as the one-reference
    multi-reference search from a zero state."""
    nx, ou, xr, nref, n = 64, 24, 2, 6, 96
    refs = synth.make_references(nref, nx, ou)
    refs[3] = refs[1]; refs[5] = refs[1]
    parts, truth = synth.make_particles(refs, n, xr, xr, 0.5, ou=ou)
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    _, cref = orc.prepare_refs(refs, mask, rg)
    d = np.zeros((n, 2), np.float32)
    params, _, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d, nthreads=8)
    won = params[:, 4].astype(int)
    copies = won[np.isin(won, (1, 3, 5))]
    assert len(copies) >= 20
    assert not (copies == 3).any()
    assert (copies == 1).any() and (copies == 5).any()
    # ormq on normalised rings against the one-reference multi-reference search
    d1 = np.zeros((n, 2), np.float32); d2 = np.zeros((n, 2), np.float32)
    pm, _, _, _ = orc.mref_iteration(parts, cref[:1], rg, xr, xr, 1.0, d1, nthreads=8)
    orc.set_ormq_normalize(True)
    try:
        po, _, _, _ = orc.reffree_iteration(parts, cref[0], rg, xr, xr, 1.0, (0, 0), d2, np.zeros((n, 6), np.float32), nthreads=8)
    finally:
        orc.set_ormq_normalize(False)
    np.testing.assert_allclose(po[:, 5], pm[:, 5], rtol=1e-6)
    assert (np.abs(d1 - d2).max(1) < 1e-6).mean() >= 0.98


@pytest.mark.parametrize("nx,ou,nref,xr", [(90, 36, 4, 3), (32, 12, 3, 2)])
def test_planted_truth_recovery(nx, ou, nref, xr):
    """particles = rot_shift2D(reference, planted); the search must return the inverse so that
    rot_shift2D(particle, found) lands back on the reference (fixes every sign convention)."""
    refs = synth.make_references(nref, nx, ou)
    n = 12
    parts, truth = synth.make_particles(refs, n, xr, xr, 0.1, ou=ou)
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    refs_n, cref = orc.prepare_refs(refs, mask, rg)
    d = np.zeros((n, 2), np.float32)
    params, infos, sums, counts = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d)
    assert (params[:, 4].astype(int) == truth["cls"]).all()
    assert (params[:, 3].astype(int) == truth["mir"]).all()
    for i in range(n):
        back = orc.rot_shift2d(parts[i], *params[i, :3], int(params[i, 3]))
        ref = refs_n[truth["cls"][i]]
        sel = mask > 0.5
        cc = np.corrcoef(back[sel], ref[sel])[0, 1]
        assert cc > 0.9, (i, cc)
        # planted centre offset: +sx without mirror, -sx with (mirror applied last)
        exp_ix = -truth["sx"][i] if truth["mir"][i] else truth["sx"][i]
        assert infos[i].ix == exp_ix and infos[i].iy == truth["sy"][i]
    assert counts.sum() == n


def test_mref_iteration_threads_equal_serial():
    refs = synth.make_references(3, 32, 12)
    parts, _ = synth.make_particles(refs, 10, 2, 2, 0.5, ou=12)
    rg = orc.rings(1, 12, 1)
    _, cref = orc.prepare_refs(refs, orc.model_circle(12, 32, 32), rg)
    d1 = np.zeros((10, 2), np.float32); d2 = d1.copy()
    p1, _, s1, c1 = orc.mref_iteration(parts, cref, rg, 2, 2, 1.0, d1, nthreads=1)
    p2, _, s2, c2 = orc.mref_iteration(parts, cref, rg, 2, 2, 1.0, d2, nthreads=4, index0=0)
    np.testing.assert_array_equal(p1, p2)
    np.testing.assert_array_equal(s1, s2)
    np.testing.assert_array_equal(c1, c2)


def test_golden_fixtures_reproduce(golden_dir):
    """the committed vectors are what the oracle computes today (guards the oracle itself)."""
    for name in ("mref_32.npz", "mref_90.npz"):
        g = np.load(os.path.join(golden_dir, name))
        ou, xr = int(g["ou"]), float(g["xr"])
        rg = orc.rings(1, ou, 1)
        _, cref = orc.prepare_refs(g["refs"], None, rg)
        np.testing.assert_allclose(cref, g["crefim"], rtol=0, atol=1e-4 * np.abs(g["crefim"]).max())
        d = np.zeros((g["particles"].shape[0], 2), np.float32)
        params, infos, sums, counts = orc.mref_iteration(g["particles"], cref, rg, xr, xr, 1.0, d)
        np.testing.assert_allclose(params, g["params0"], rtol=1e-5, atol=1e-4)
        assert [infos[i].jtot for i in range(len(d))] == list(g["jtot0"])
        np.testing.assert_array_equal(counts, g["counts0"])


def test_reffree_iteration_ormq_consistency(golden_dir):
    g = np.load(os.path.join(golden_dir, "reffree_32.npz"))
    ou, xr = int(g["ou"]), float(g["xr"])
    rg = orc.rings(1, ou, 1)
    _, cref = orc.prepare_refs(g["tavg"], None, rg)
    n = g["particles"].shape[0]
    d = np.zeros((n, 2), np.float32); params = np.zeros((n, 6), np.float32)
    params, infos, sums, ss = orc.reffree_iteration(g["particles"], cref[0], rg, xr, xr, 1.0, (0, 0), d, params)
    np.testing.assert_allclose(params, g["params0"], rtol=1e-5, atol=1e-4)
    # ormq directly on particle 0 gives the same answer
    out, info = orc.ormq(g["particles"][0], cref[0], [xr, xr], [xr, xr], 1.0, rg, 17.0, 17.0)
    assert out[0] == pytest.approx(params[0, 0], abs=1e-3) and int(out[3]) == int(params[0, 3])
    sx_sum = sum((p[1] if p[3] == 0 else -p[1]) for p in params)
    assert ss[0] == pytest.approx(sx_sum, rel=1e-5, abs=1e-5)


def test_config0_plumbing_six_iterations_class_sizes_converge():
    """BASELINE configs[0]: the CPU path alone -- 1000 synthetic 90 x 90 particles, nref = 5, xr = yr = 3, ou = 36,
    maxit = 6 -- as the loop of mref_ali2d (test_mref_gpu_align.py:1005-1060, 517-575 without a user function): references
    -> Polar2Dm / Frngs / Applyws, state round trip through the float32 header values, search, rot_shift2D + even/odd
    sums, (even + odd) / n, normalize.mask.  Started from references blurred by averaging a random fifth of the stack
    per class, the class sizes settle on the planted ones and stop changing."""
    from cryo_ralib_amd import synth
    nx, ou, nref, xr, n, maxit = 90, 36, 5, 3, 1000, 6
    refs = synth.make_references(nref, nx, ou)
    parts, truth = synth.make_particles(refs, n, xr, xr, 1.0, ou=ou)
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    # initial references: the planted ones + noise of the same level as the particles (a poor start)
    rng = np.random.default_rng(3)
    cur = np.stack([orc.normalize_mask(r + rng.standard_normal((nx, nx)).astype(np.float32), mask, 1) for r in refs])
    d = np.zeros((n, 2), np.float32)
    params = None
    sizes, moved = [], []
    prev_assign = None
    for it in range(maxit):
        _, cref = orc.prepare_refs(cur, None, rg)
        if params is not None:
            d = orc.state_from_params(params, 0)
        params, infos, sums, counts = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d, nthreads=8)
        assert counts.sum() == n and (counts >= 4).all()
        assign = params[:, 4].astype(int)
        if prev_assign is not None:
            moved.append(int((assign != prev_assign).sum()))
        prev_assign = assign
        sizes.append(counts.copy())
        cur = np.stack([orc.normalize_mask((sums[j, 0] + sums[j, 1]) * np.float32(1.0 / float(counts[j])), mask, 1)
                        for j in range(nref)])
    planted = np.bincount(truth["cls"], minlength=nref)
    # convergence of the class sizes: the last two iterations agree, and with the planted sizes, to a few particles
    assert np.abs(sizes[-1] - sizes[-2]).max() <= 5, sizes
    assert np.abs(sizes[-1] - planted).max() <= 0.03 * n, (sizes[-1], planted)
    assert moved[-1] <= moved[0] and moved[-1] <= 0.02 * n, moved
    assert (prev_assign == truth["cls"]).mean() > 0.97
