"""GPU parity tests of the particle-resident search kernels for boxes that do not fit the four ring buffers of the 90 x 90 kernels:
rings of 512 samples (search_solo_kernel / search_duo_kernel, ralign_solo.h / ralign_duo.h: ou = 41 .. ~62, the geometry class of the
reference's own documented run, notebook/00_Multireference_Alignment.ipynb cell 3: 130 x 130, ou = 52, nref = 50) and rings of 256
samples in boxes of ~96 .. 150 pixels (search_pair_kernel, ralign_pair.h).  Same bars as tests/test_gpu_parity.py: identical integer
assignments (no tie allowance), CCF peaks within 1e-4."""
import os

import numpy as np
import pytest
import torch

from cryo_ralib_amd import api, synth
from cryo_ralib_amd.mref import MrefAligner
from oracle import oracle as orc

from test_gpu_parity import (_log_flips, assert_images_close, compare_search, default_path_only, oracle_setup, polar_stage_check,
                             run_engine)

pytestmark = pytest.mark.gpu

SOLO = 3        # ra_search_path: particle-resident, one offset per pass


@pytest.mark.parametrize("nx,ou,xr,mode", [(130, 52, 3, api.RA_MODE_MREF), (128, 60, 3, api.RA_MODE_MREF),
                                           (128, 60, 3, api.RA_MODE_REFFREE), (101, 44, 2, api.RA_MODE_MREF),
                                           (100, 40, 3, api.RA_MODE_MREF), (128, 40, 4, api.RA_MODE_REFFREE)])      # the last two: search_pair_kernel (ou = 40: the
                                                                                                      # one radius the four-offset kernels do not fit)
def test_solo_polar_stage_bin_for_bin(nx, ou, xr, mode):
    """Polar2Dm -> (Normalize_ring) -> Frngs of every in-window search offset through the ring jobs of the solo / pair kernels (the
    512-sample job included), element by element against the oracle in EMAN2's packed ring layout"""
    default_path_only("RALIGN_SOLO", "RALIGN_GENERIC", "RALIGN_PAIR")
    eng = api.Engine(nx, ou, xr, xr, 1.0, 2 if mode == api.RA_MODE_MREF else 1, mode)
    assert eng.search_path == SOLO and eng.maxrin == (512 if ou > 40 else 256)
    eng.close()
    polar_stage_check(nx, ou, xr, mode)


def test_two_offsets_per_pass_equal_one_offset_per_pass_bit_for_bit(monkeypatch):
    """search_duo_kernel (the default: two offsets per pass through the one ring buffer, the first offset's slice held in registers)
    against search_solo_kernel with the same ring jobs (RALIGN_DUO=0 RALIGN_SOLO_JOBS=1): the same spectra, the same chain of
    matrix instructions per block row, the same transforms -- every record equal to the bit, odd offset counts and edge-limited
    windows included"""
    default_path_only("RALIGN_SOLO", "RALIGN_GENERIC", "RALIGN_DUO")
    nx, ou, xr = 130, 52, 3
    for nref, n in ((50, 64), (10, 96), (3, 48)):
        refs = synth.make_references(nref, nx, ou)
        parts, _ = synth.make_particles(refs, n, xr, xr, 1.0, ou=ou)
        rg, mask, refs_n, cref = oracle_setup(refs, ou, nx)
        st0 = np.zeros((n, 2), np.float32)
        st0[::4] = (11, -12); st0[1::7] = (-12, 3)          # windows cut at the edge: odd numbers of live offsets
        out = []
        for duo in ("1", "0"):
            monkeypatch.setenv("RALIGN_DUO", duo)
            monkeypatch.setenv("RALIGN_SOLO_JOBS", "1")
            eng, tp, st, res = run_engine(parts, refs_n, ou, xr, xr, 1.0, state=st0)
            assert eng.search_path == SOLO
            out.append((api.Engine.result_to_numpy(res).copy(), st.cpu().numpy().copy()))
            eng.close()
        for f in ("ref_id", "mirror", "angle_bin", "shift_idx", "peak", "alpha", "sx", "sy"):
            np.testing.assert_array_equal(out[0][0][f], out[1][0][f], err_msg=f)
        np.testing.assert_array_equal(out[0][1], out[1][1])


def test_solo_light_ring_jobs_bin_for_bin(monkeypatch):
    """RALIGN_SOLO_JOBS=1: the 8 x 8 x 4 job for 512-sample rings (ring_job512, 32 lanes per ring) and the 16-lane job for
    256-sample rings -- kept as a measured alternative (slower: the ring jobs are bound by the LDS array)"""
    default_path_only("RALIGN_SOLO", "RALIGN_GENERIC")
    monkeypatch.setenv("RALIGN_SOLO_JOBS", "1")
    polar_stage_check(128, 60, 3, api.RA_MODE_MREF)
    polar_stage_check(130, 52, 3, api.RA_MODE_REFFREE)


def test_generic_polar_stage_at_maxrin_512_stays_covered(monkeypatch):
    """the size-generic polar kernel at a geometry the solo kernel took over (RALIGN_SOLO=0)"""
    monkeypatch.setenv("RALIGN_SOLO", "0")
    eng = api.Engine(128, 50, 2, 2, 1.0, 2)
    assert eng.search_path == 2
    eng.close()
    polar_stage_check(128, 50, 2, api.RA_MODE_MREF, n=2, rtol=2e-5)


def crop4_expected(ou, nref):
    """engines of the size-generic class whose rings end at 256 samples and fit four ring buffers next to a CROP of the image run the
    kernels of the 90 x 90 boxes (search_fused_kernel up to 14 references, search_tiled_kernel beyond) instead of the pair kernel"""
    off = any(os.environ.get(sw) == "0" for sw in ("RALIGN_TCROP", "RALIGN_FUSED", "RALIGN_CROP")) or (nref >= 15 and os.environ.get("RALIGN_TILED") == "0")
    return ou <= 39 and not off          # (ou = 37 .. 39: the fused kernel only, 38 and 39 with the rings 4 floats apart)


def needs_crop(nx):
    """boxes the solo / duo / pair kernels reach only because their LDS image is a crop: skipped under RALIGN_CROP=0"""
    if nx > 150:
        default_path_only("RALIGN_CROP")


@pytest.mark.parametrize("nx,ou,nref,n,sigma", [(130, 52, 50, 160, 0.25), (130, 52, 50, 160, 1.0),       # the reference notebook's geometry
                                                (128, 60, 10, 384, 0.25), (128, 60, 10, 384, 1.0),
                                                (128, 60, 7, 96, 1.0),                                   # odd reference count: a half-filled pair
                                                (130, 52, 12, 96, 1.0),                                  # two tiles of three pairs
                                                (131, 58, 3, 64, 0.5),                                   # odd box
                                                (100, 40, 10, 384, 1.0), (128, 40, 50, 128, 1.0),        # search_pair_kernel: one tile, five tiles
                                                (101, 37, 7, 96, 0.25), (144, 40, 1, 64, 1.0),           # odd box and reference count; one reference
                                                (96, 36, 10, 128, 1.0), (112, 36, 24, 96, 0.5),          # search_tiled_kernel on a crop (ou <= 36, 7 and more references)
                                                (200, 40, 10, 64, 1.0), (192, 56, 6, 48, 1.0)])          # boxes far larger than the rings: cropped LDS image
def test_solo_search_against_oracle(nx, ou, nref, n, sigma):
    default_path_only("RALIGN_SOLO", "RALIGN_GENERIC", "RALIGN_PAIR")
    needs_crop(nx)
    xr = 3
    refs = synth.make_references(nref, nx, ou)
    parts, truth = synth.make_particles(refs, n, xr, xr, sigma, ou=ou)
    rg, mask, refs_n, cref = oracle_setup(refs, ou, nx)
    d = np.zeros((n, 2), np.float32)
    params, infos, sums, counts = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d, nthreads=16)
    eng, tp, st, res = run_engine(parts, refs_n, ou, xr, xr, 1.0)
    # four offsets per pass: the fused / tiled kernel on a crop of the image (unless a switch takes it away: then the pair kernel)
    tcrop = crop4_expected(ou, nref)
    assert eng.search_path == (1 if tcrop else SOLO) and bool(eng.search_tiled) == (tcrop and nref >= 15)
    r = eng.result_to_numpy(res)
    flips = compare_search(r, st.cpu().numpy(), params, infos, d)
    _log_flips("solo %d/%d nref=%d sigma=%g" % (nx, ou, nref, sigma), n, flips)
    assert flips == 0
    if sigma == 0.25:
        assert (r["ref_id"] == truth["cls"]).all() and (r["mirror"] == truth["mir"]).all()
    gs = torch.zeros((nref, 2, nx, nx), device=eng.dev)
    gc = torch.zeros(nref, dtype=torch.int32, device=eng.dev)
    eng.transform_accumulate(tp, res, 0, None, gs, gc)
    eng.sync()
    np.testing.assert_array_equal(gc.cpu().numpy(), counts)
    if sigma <= 0.5:      # (at sigma = 1 a 1e-5 degree difference of the sub-bin angle moves single pixels of a 7-image sum by O(sigma):
        assert_images_close(gs.cpu().numpy(), sums, mask, 2e-5 * np.abs(sums).max() + 1e-4)      # quadri's cell borders, see assert_images_close)
    eng.close()


@pytest.mark.parametrize("nx,ou,switch", [(128, 60, "RALIGN_SOLO"), (100, 40, "RALIGN_PAIR")])
def test_solo_search_equals_generic_search(monkeypatch, nx, ou, switch):
    """same inputs through the solo / duo / pair kernel and through the size-generic kernels it replaces: identical integer assignments"""
    default_path_only("RALIGN_SOLO", "RALIGN_GENERIC", "RALIGN_PAIR")
    nref, xr, n = 10, 3, 96
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 1.0, ou=ou)
    rg, mask, refs_n, cref = oracle_setup(refs, ou, nx)
    st0 = np.zeros((n, 2), np.float32)
    m = nx // 2 + 1 - ou - 2
    st0[::3] = (m - 1, -1); st0[1::5] = (-m, m)      # edge-limited windows: the resident kernels skip their out-of-window offsets
    eng, tp, st, res = run_engine(parts, refs_n, ou, xr, xr, 1.0, state=st0)
    assert eng.search_path == SOLO
    solo = api.Engine.result_to_numpy(res).copy()
    st_solo = st.cpu().numpy().copy()
    eng.close()
    monkeypatch.setenv(switch, "0")
    eng, tp, st, res = run_engine(parts, refs_n, ou, xr, xr, 1.0, state=st0)
    assert eng.search_path == 2
    gen = api.Engine.result_to_numpy(res).copy()
    for f in ("ref_id", "mirror", "angle_bin", "shift_idx"):
        assert (solo[f] == gen[f]).all(), f
    assert (np.abs(solo["peak"] - gen["peak"]) / np.abs(solo["peak"])).max() < 1e-5
    np.testing.assert_array_equal(st_solo, st.cpu().numpy())
    eng.close()


@pytest.mark.parametrize("nx,ou", [(130, 52), (100, 40),
                                   (200, 40), (256, 36), (192, 56), (161, 45)])      # the LDS image is a crop that follows the particle's centre
def test_solo_edge_limited_windows_and_reset_rule(nx, ou):
    """accumulated shifts at and beyond the edge of the box: search_range cuts the window, |shift| > mashi resets it
    (test_mref_gpu_align.py:1030-1038); the solo / duo / pair kernels never sample an out-of-window offset"""
    default_path_only("RALIGN_SOLO", "RALIGN_GENERIC", "RALIGN_PAIR")
    needs_crop(nx)
    nref, xr, n = 4, 3, 40
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 0.5, ou=ou)
    rg, mask, refs_n, cref = oracle_setup(refs, ou, nx)
    rng = np.random.default_rng(5)
    mashi = nx // 2 + 1 - ou - 2
    d0 = rng.integers(-mashi - 2, mashi + 3, size=(n, 2)).astype(np.float32)
    d0[:4] = [(mashi, mashi), (-mashi, mashi), (mashi + 1, 0), (0, -mashi - 1)]
    d = d0.copy()
    params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d, nthreads=8)
    eng, tp, st, res = run_engine(parts, refs_n, ou, xr, xr, 1.0, state=d0)
    assert eng.search_path == (1 if (nx > 150 and crop4_expected(ou, nref)) else SOLO)
    flips = compare_search(eng.result_to_numpy(res), st.cpu().numpy(), params, infos, d)
    _log_flips("solo edge-limited windows", n, flips)
    eng.close()


@pytest.mark.parametrize("nx,ou", [(128, 56), (112, 40), (144, 32)])      # duo, pair, fused kernel on a crop
def test_solo_reference_free_and_nomirror(nx, ou):
    """ormq's rules (no Normalize_ring, one reference) and its nomirror form (Crosrng_ns) at maxrin 512 (duo kernel) and at
    maxrin 256 in a large box (pair kernel)"""
    default_path_only("RALIGN_SOLO", "RALIGN_GENERIC", "RALIGN_PAIR")
    xr, n = 3, 96
    refs = synth.make_references(1, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 1.0, ou=ou)
    rg = orc.rings(1, ou, 1)
    refs_n, cref = orc.prepare_refs(refs, None, rg)
    for nomirror in (False, True):
        orc.set_nomirror(nomirror)
        try:
            d = np.zeros((n, 2), np.float32)
            params, infos, _, _ = orc.reffree_iteration(parts, cref[0], rg, xr, xr, 1.0, (0, 0), d, np.zeros((n, 6), np.float32), nthreads=8)
        finally:
            orc.set_nomirror(False)
        eng = api.Engine(nx, ou, xr, xr, 1.0, 1, api.RA_MODE_REFFREE)
        assert eng.search_path == (1 if crop4_expected(ou, 1) else SOLO)
        eng.set_nomirror(nomirror)
        eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
        st, res = eng.new_state(n), eng.new_result(n)
        eng.align(torch.from_numpy(parts).to(eng.dev), st, res)
        eng.sync()
        r = eng.result_to_numpy(res)
        if nomirror:
            assert (r["mirror"] == 0).all()
        flips = compare_search(r, st.cpu().numpy(), params, infos, d)
        _log_flips("solo reffree nomirror=%d" % nomirror, n, flips)
        eng.close()


@pytest.mark.parametrize("nx,ou", [(130, 52), (176, 40), (160, 34)])      # cropped LDS image, class sums by output tiles: pair kernel; fused kernel
def test_solo_in_the_iteration_loop(nx, ou):
    """three mref_ali2d iterations of the host driver at the notebook's geometry (search, rot_shift2D + class sums, reference
    update, state round trip through the header values) against the same loop built from oracle calls"""
    default_path_only("RALIGN_SOLO", "RALIGN_GENERIC", "RALIGN_PAIR")
    from test_gpu_parity import _oracle_mref_loop_step, assert_alpha_equal_to_the_ulp
    needs_crop(nx)
    nref, xr, n = 4, 3, 96
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 0.5, ou=ou)
    al = MrefAligner(parts, refs, ou, xr, xr, 1.0, preprocess=True, refine=-1)
    assert al.engine.search_path == (1 if (nx > 150 and crop4_expected(ou, nref)) else SOLO)
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    cur = np.stack([orc.normalize_mask(r, mask, 1) for r in refs])
    op = np.stack([orc.normalize_mask(p, mask, 0) for p in parts])
    d = np.zeros((n, 2), np.float32)
    prev = None
    for it in range(3):
        params, infos, sums, counts, d = _oracle_mref_loop_step(op, cur, rg, mask, xr, prev, d, True)
        got_counts = al.iterate()
        r = al.params()
        flips = compare_search(r, al.state.cpu().numpy(), params, infos, d)
        _log_flips("solo mref loop it=%d" % it, n, flips)
        prev = params
        np.testing.assert_array_equal(got_counts, counts)
        assert_alpha_equal_to_the_ulp(r["alpha"], params[:, 0])
        cur = np.stack([orc.normalize_mask((sums[j, 0] + sums[j, 1]) * np.float32(1.0 / float(counts[j])), mask, 1)
                        for j in range(nref)])
        # (sx / sy agree to one float32 ulp -- the device's double sin / cos against libm's --, which moves a pixel at one of quadri's
        # cell borders now and then: a handful of the 67 600 pixels may differ by more than the rounding of the sums)
        diff = np.abs(al.refs.cpu().numpy() - cur)
        assert (diff > 2e-6 * np.abs(cur).max()).sum() <= 4 and diff.max() < 1e-3 * np.abs(cur).max(), (diff.max(), (diff > 2e-6 * np.abs(cur).max()).sum())
    al.close()


@pytest.mark.parametrize("nx,ou", [(104, 40), (128, 52)])
def test_multi_stage_schedule_with_fractional_steps(nx, ou):
    """--xr "2 1" --ts "1 0.5" (reset_shifts between the stages, a half-pixel grid in the second) through the pair / duo kernels:
    fractional sampling centres, windows from search_range with a fractional step, every iteration against ali2d_single_iter"""
    from cryo_ralib_amd.mref import RefFreeAligner
    default_path_only("RALIGN_SOLO", "RALIGN_GENERIC", "RALIGN_PAIR")
    n = 96
    refs = synth.make_references(1, nx, ou)
    parts, _ = synth.make_particles(refs, n, 2, 2, 0.5, ou=ou)
    rg = orc.rings(1, ou, 1)
    al = RefFreeAligner(parts, ou, "2 1", "-1", "1 0.5")
    assert al.engine.search_path == SOLO and al.engine.num_shifts == 25
    for stage in (0, 1):
        al.set_stage(stage)
        x, y, t = al.stages[stage]
        for it in range(2):
            prev = al.params().copy()
            d = al.state.cpu().numpy().copy()
            al.iterate(0, None)
            al.engine.sync()
            _, cref = orc.prepare_refs(al.tavg.cpu().numpy(), None, rg)
            params = np.zeros((n, 6), np.float32)
            params[:, 0] = prev["alpha"]; params[:, 1] = prev["sx"]; params[:, 2] = prev["sy"]; params[:, 3] = prev["mirror"]
            params, infos, _, _ = orc.reffree_iteration(parts, cref[0], rg, x, y, t, (0, 0), d, params, nthreads=16)
            flips = compare_search(al.params(), al.state.cpu().numpy(), params, infos, d)
            _log_flips("%d/%d stage %d it %d" % (nx, ou, stage, it), n, flips)
    al.close()


def test_reset_shifts_to_a_wider_range_replans_the_cropped_image():
    """ra_reset_shifts at a constant offset count but a wider pixel range (xr = 1, ts = 0.5 -> xr = 4, ts = 2: 25 offsets both): the crop
    of the image the pair kernel keeps in LDS is sized by the range, so the engine plans it again -- results equal those of an engine
    created with the wide range, and the oracle's"""
    default_path_only("RALIGN_SOLO", "RALIGN_GENERIC", "RALIGN_PAIR", "RALIGN_CROP")
    nx, ou, nref, n = 176, 40, 3, 48
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, 4, 4, 0.5, ou=ou)
    rg, mask, refs_n, cref = oracle_setup(refs, ou, nx)
    d = np.zeros((n, 2), np.float32)
    params, infos, _, _ = orc.mref_iteration(parts, cref, rg, 4, 4, 2.0, d, nthreads=8)
    outs = []
    for first in ((1, 1, 0.5), (4, 4, 2.0)):
        eng = api.Engine(nx, ou, first[0], first[1], first[2], nref, api.RA_MODE_MREF)
        assert eng.search_path == SOLO
        eng.reset_shifts(4, 4, 2.0)
        assert eng.search_path == SOLO
        eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
        st, res = eng.new_state(n), eng.new_result(n)
        eng.align(torch.from_numpy(parts).to(eng.dev), st, res)
        eng.sync()
        outs.append((eng.result_to_numpy(res).copy(), st.cpu().numpy().copy()))
        eng.close()
    for f in api.RESULT_DTYPE.names:
        np.testing.assert_array_equal(outs[0][0][f], outs[1][0][f], err_msg=f)
    flips = compare_search(outs[0][0], outs[0][1], params, infos, d)
    _log_flips("pair kernel after reset_shifts to a wider range", n, flips)


@pytest.mark.parametrize("nx,ou,nref", [(96, 36, 10), (112, 36, 24)])
def test_pair_kernel_with_grown_ring_buffers(nx, ou, nref, monkeypatch):
    """RALIGN_TCROP=0: the geometries search_tiled_kernel takes on a crop of the image go through search_pair_kernel, whose ring
    buffers grow to hold the 12 spectra of a six-pair tile (at ou = 36 the rings of an offset are shorter than that)"""
    default_path_only("RALIGN_SOLO", "RALIGN_GENERIC", "RALIGN_PAIR")
    monkeypatch.setenv("RALIGN_TCROP", "0")
    xr, n = 3, 96
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 1.0, ou=ou)
    rg, mask, refs_n, cref = oracle_setup(refs, ou, nx)
    d = np.zeros((n, 2), np.float32)
    params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d, nthreads=16)
    eng, tp, st, res = run_engine(parts, refs_n, ou, xr, xr, 1.0)
    assert eng.search_path == SOLO and eng.search_offsets_per_pass == 2
    flips = compare_search(eng.result_to_numpy(res), st.cpu().numpy(), params, infos, d)
    _log_flips("pair kernel, grown ring buffers %d/%d nref=%d" % (nx, ou, nref), n, flips)
    eng.close()


@pytest.mark.parametrize("nx,ou,nref,xr,ts", [(256, 36, 10, 3, 1.0), (160, 30, 24, 2, 0.5), (200, 34, 50, 3, 1.0), (141, 36, 7, 3, 1.0),
                                              (176, 33, 2, 3, 1.0), (256, 30, 14, 2, 0.5),
                                              (128, 38, 10, 3, 1.0), (150, 39, 3, 2, 1.0)])      # rings 4 floats apart: ou = 38, 39 fit too
def test_tiled_kernel_on_a_cropped_image(nx, ou, nref, xr, ts):
    """boxes far larger than rings of 256 samples (engines of the size-generic class): search_fused_kernel (up to 14 references) /
    search_tiled_kernel, four offsets per pass, over a crop of the image that follows the particle's centre WITHOUT being clamped to the box (the kernel samples every offset of the
    window; those search_range excludes read whatever the crop holds there and are never looked at).  Shifts up to and beyond mashi,
    the dense offset stream with partial last passes, a half-pixel grid, 1 - 5 reference tiles"""
    default_path_only("RALIGN_SOLO", "RALIGN_GENERIC", "RALIGN_PAIR", "RALIGN_CROP", "RALIGN_TCROP", "RALIGN_FUSED", "RALIGN_TILED", "RALIGN_TIGHT_RINGS")
    n = 61
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 0.5, ou=ou)
    rg, mask, refs_n, cref = oracle_setup(refs, ou, nx)
    rng = np.random.default_rng(9)
    mashi = nx // 2 + 1 - ou - 2
    d0 = np.zeros((n, 2), np.float32)
    d0[:24] = rng.integers(-mashi - 2, mashi + 3, size=(24, 2)).astype(np.float32)
    d0[:4] = [(mashi, mashi), (-mashi, mashi), (mashi + 1, 0), (0, -mashi - 1)]
    d = d0.copy()
    params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, ts, d, nthreads=16)
    os.environ["RALIGN_GRID"] = "7"          # workgroups walk over several particles: passes that hold offsets of two
    try:
        eng, tp, st, res = run_engine(parts, refs_n, ou, xr, xr, ts, state=d0)
    finally:
        os.environ.pop("RALIGN_GRID", None)
    assert eng.search_path == 1 and bool(eng.search_tiled) == (nref >= 15)
    flips = compare_search(eng.result_to_numpy(res), st.cpu().numpy(), params, infos, d)
    _log_flips("fused / tiled kernel on a crop %d/%d nref=%d ts=%g" % (nx, ou, nref, ts), n, flips)
    eng.close()


def test_tiled_kernel_on_a_crop_follows_reset_shifts():
    """ra_reset_shifts on an engine whose search runs over a crop of the image: a narrower range keeps the crop (results equal a
    fresh engine's), a wider one at the same offset count is refused -- the crop was sized for the range of ra_create, like the
    image border of the 90 x 90 kernels"""
    default_path_only("RALIGN_SOLO", "RALIGN_GENERIC", "RALIGN_PAIR", "RALIGN_CROP", "RALIGN_TCROP", "RALIGN_FUSED", "RALIGN_TILED")
    nx, ou, nref, n = 176, 36, 10, 40
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, 1, 1, 0.5, ou=ou)
    rg, mask, refs_n, cref = oracle_setup(refs, ou, nx)
    outs = []
    for first in ((2, 2, 1.0), (1, 1, 0.5)):
        eng = api.Engine(nx, ou, first[0], first[1], first[2], nref, api.RA_MODE_MREF)
        assert eng.search_path == 1
        eng.reset_shifts(1, 1, 0.5)
        eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
        st, res = eng.new_state(n), eng.new_result(n)
        eng.align(torch.from_numpy(parts).to(eng.dev), st, res)
        eng.sync()
        outs.append((eng.result_to_numpy(res).copy(), st.cpu().numpy().copy()))
        if first[0] == 1:
            with pytest.raises(api.EngineError):
                eng.reset_shifts(4, 4, 2.0)
        eng.close()
    for f in api.RESULT_DTYPE.names:
        np.testing.assert_array_equal(outs[0][0][f], outs[1][0][f], err_msg=f)
    d = np.zeros((n, 2), np.float32)
    params, infos, _, _ = orc.mref_iteration(parts, cref, rg, 1, 1, 0.5, d, nthreads=8)
    flips = compare_search(outs[0][0], outs[0][1], params, infos, d)
    _log_flips("tiled kernel on a crop after reset_shifts", n, flips)
