"""GPU parity tests of the two engine options that hedge the unpinned choices of the EMAN2 CPU path (include/ralign.h:
ra_options; SURVEY.md section 7 "Hard parts", Appendix A.3 / A.4; reference call sites test_mref_gpu_align.py:1015, 1043-1044):

  * Normalize_ring on / off independent of the mode -- through EVERY kernel family (fused, tiled, kernel pair, pair, duo,
    size-generic), against the oracle's `normalize_ring` flag;
  * Util::alrl_ms with Util::quadri instead of Util::bilinear -- through the size-generic kernels and the exact
    re-evaluation, against the oracle's ORC_INTERP_QUADRI.

The bar is the one of tests/test_gpu_parity.py: identical integer assignments, CCF peaks within 1e-4.
"""
import numpy as np
import pytest
import torch

from cryo_ralib_amd import api, geometry, synth
from oracle import oracle as orc
from test_gpu_parity import compare_search, assert_alpha_equal_to_the_ulp, default_path_only, _log_flips

pytestmark = pytest.mark.gpu


def _search(eng, parts, refs_n, state=None):
    n = parts.shape[0]
    eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
    tp = torch.from_numpy(parts).to(eng.dev)
    st = eng.new_state(n) if state is None else torch.from_numpy(state.copy()).to(eng.dev)
    res = eng.new_result(n)
    eng.align(tp, st, res)
    eng.sync()
    return api.Engine.result_to_numpy(res), st.cpu().numpy()


# geometry -> the kernel family the default dispatch picks for it (ra_search_path, tiled flag)
FAMILIES = [
    # nx, ou, xr, nref, n, env, (path, tiled)
    (90, 36, 3, 10, 48, {}, (1, False)),                        # search_fused_kernel (BASELINE configs[1] geometry)
    (32, 12, 2, 3, 48, {}, (1, False)),                         # ... on 128-sample rings
    (90, 36, 3, 20, 32, {}, (1, True)),                         # search_tiled_kernel
    (90, 36, 3, 10, 32, {"RALIGN_FUSED": "0"}, (0, False)),     # polar_fft_kernel + ccf_kernel
    (100, 40, 3, 4, 24, {}, (3, False)),                        # search_pair_kernel
    (128, 60, 2, 4, 12, {}, (3, False)),                        # search_duo_kernel
    (128, 36, 3, 6, 24, {}, (1, False)),                        # search_fused_kernel on a crop of the image
    (90, 36, 3, 10, 24, {"RALIGN_GENERIC": "1"}, (2, False)),   # size-generic kernels
]
FAMILY_IDS = ["fused-90-36", "fused-32-12", "tiled", "kernel-pair", "pair-100-40", "duo-128-60", "fused-crop-128-36", "generic"]


@pytest.mark.parametrize("nx,ou,xr,nref,n,env,want", FAMILIES, ids=FAMILY_IDS)
def test_normalize_ring_off_in_multireference_mode(nx, ou, xr, nref, n, env, want, monkeypatch):
    """RA_MODE_MREF with normalize_ring = 0: Util.multiref_polar_ali_2d WITHOUT its Normalize_ring call (the oracle's flag) --
    window rule, reference loop and tie order stay the mode's"""
    default_path_only("RALIGN_FUSED", "RALIGN_TILED", "RALIGN_GENERIC", "RALIGN_PAIR", "RALIGN_SOLO", "RALIGN_DUO", "RALIGN_TCROP", "RALIGN_CROP")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 1.0, ou=ou)
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    refs_n, cref = orc.prepare_refs(refs, mask, rg)
    d = np.zeros((n, 2), np.float32)
    d[1] = (1, -1)
    eng = api.Engine(nx, ou, xr, xr, 1.0, nref, api.RA_MODE_MREF, normalize_ring=False)
    assert (eng.search_path, eng.search_tiled) == want, (eng.search_path, eng.search_tiled)
    assert eng.options == (api.RA_INTERP_BILINEAR, 0)
    d_off, d_on = d.copy(), d.copy()          # the oracle's loops advance the state in place
    want_off = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d_off, nthreads=8, normalize=False)
    want_on = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d_on, nthreads=8, normalize=True)
    # the option matters on this input: without the normalisation the peaks carry the particle's own scale
    assert np.abs(want_off[0][:, 5] - want_on[0][:, 5]).max() > 1e-3 * np.abs(want_on[0][:, 5]).max()
    r, st = _search(eng, parts, refs_n, d)
    flips = compare_search(r, st, want_off[0], want_off[1], d_off)
    _log_flips("normalize_ring off / mref / %d-%d-%d %s" % (nx, ou, nref, env or ""), n, flips)
    # ... and back to the mode's default at run time (ra_set_normalize_ring(-1)): the plain multi-reference search
    eng.set_normalize_ring(None)
    assert eng.options[1] == 1
    r, st = _search(eng, parts, refs_n, d)
    compare_search(r, st, want_on[0], want_on[1], d_on)
    eng.close()


@pytest.mark.parametrize("nx,ou,xr,env,want", [(90, 36, 3, {}, 1), (32, 12, 2, {}, 1), (100, 40, 3, {}, 3),
                                               (90, 36, 3, {"RALIGN_GENERIC": "1"}, 2)],
                         ids=["fused-90-36", "fused-32-12", "pair-100-40", "generic"])
def test_normalize_ring_on_in_reference_free_mode(nx, ou, xr, env, want, monkeypatch):
    """RA_MODE_REFFREE with normalize_ring = 1: ormq's single-reference search on NORMALISED rings -- the checker's ormq with
    Normalize_ring between Polar2Dm and Frngs (orc.set_ormq_normalize).  From a zero state the windows of ormq (clamp) and of
    multiref_polar_ali_2d (reset) coincide, so the one-reference multi-reference search finds the same peaks (it may order an
    exact float tie of two offsets differently: its running peak is a float, ormq's a double)"""
    default_path_only("RALIGN_FUSED", "RALIGN_TILED", "RALIGN_GENERIC", "RALIGN_PAIR", "RALIGN_TCROP", "RALIGN_CROP")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n = 40
    refs = synth.make_references(1, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 1.0, ou=ou)
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    refs_n, cref = orc.prepare_refs(refs, mask, rg)
    d = np.zeros((n, 2), np.float32)
    eng = api.Engine(nx, ou, xr, xr, 1.0, 1, api.RA_MODE_REFFREE, normalize_ring=True)
    assert eng.search_path == want
    assert eng.options == (api.RA_INTERP_BILINEAR, 1)
    d1 = d.copy()
    orc.set_ormq_normalize(True)
    try:
        params, infos, _, _ = orc.reffree_iteration(parts, cref[0], rg, xr, xr, 1.0, (0, 0), d1, np.zeros((n, 6), np.float32), nthreads=8)
    finally:
        orc.set_ormq_normalize(False)
    r, st = _search(eng, parts, refs_n)
    flips = compare_search(r, st, params, infos, d1)
    _log_flips("normalize_ring on / reffree / %d-%d %s" % (nx, ou, env or ""), n, flips)
    dm = d.copy()
    pm, _, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, dm, nthreads=8, normalize=True)
    np.testing.assert_allclose(r["peak"], pm[:, 5], rtol=1e-4)
    # switched off at run time: plain ormq
    eng.set_normalize_ring(False)
    p0 = np.zeros((n, 6), np.float32)
    d2 = d.copy()
    params, infos, _, _ = orc.reffree_iteration(parts, cref[0], rg, xr, xr, 1.0, (0, 0), d2, p0, nthreads=8)
    r, st = _search(eng, parts, refs_n)
    compare_search(r, st, params, infos, d2)
    eng.close()


@pytest.mark.parametrize("nx,ou,xr,nref,mode", [(90, 36, 3, 10, api.RA_MODE_MREF), (32, 12, 2, 3, api.RA_MODE_MREF),
                                                (90, 36, 3, 1, api.RA_MODE_REFFREE), (32, 12, 2, 1, api.RA_MODE_REFFREE),
                                                (128, 60, 2, 3, api.RA_MODE_MREF)],
                         ids=["mref-90-36-R10", "mref-32-12-R3", "reffree-90-36", "reffree-32-12", "mref-128-60-R3"])
def test_quadri_interpolation_against_the_oracle(nx, ou, xr, nref, mode):
    """RA_INTERP_QUADRI: Polar2Dm / alrl_ms with Util::quadri (older EMAN2 releases) on particles AND references -- prepared
    references, the search (size-generic kernels whatever the geometry) and the exact re-evaluation of every winner"""
    n = 32 if nx < 128 else 10
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 1.0, ou=ou)
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    refs_n, cref = orc.prepare_refs(refs, mask, rg, interp=orc.INTERP_QUADRI)
    _, cref_bil = orc.prepare_refs(refs, mask, rg)
    assert np.abs(cref - cref_bil).max() > 1e-4 * np.abs(cref).max()       # the option matters on this input
    d = np.zeros((n, 2), np.float32)
    d[2] = (-1, 1)
    d_out = d.copy()
    if mode == api.RA_MODE_MREF:
        params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d_out, nthreads=8, interp=orc.INTERP_QUADRI)
    else:
        p0 = np.zeros((n, 6), np.float32)
        params, infos, _, _ = orc.reffree_iteration(parts, cref[0], rg, xr, xr, 1.0, (0, 0), d_out, p0, nthreads=8,
                                                    interp=orc.INTERP_QUADRI)
    eng = api.Engine(nx, ou, xr, xr, 1.0, nref, mode, interp=api.RA_INTERP_QUADRI)
    assert eng.search_path == 2, "quadri runs through the size-generic kernels"
    assert eng.options[0] == api.RA_INTERP_QUADRI
    eng.set_refine(-1.0)           # every winner through refine_winner_kernel: the sub-bin angle in the CPU path's arithmetic
    eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
    got = eng.prepared_references()
    assert np.abs(got - cref).max() < 1e-6 * np.abs(cref).max()
    r, st = _search(eng, parts, refs_n, d)
    flips = compare_search(r, st, params, infos, d_out)
    assert_alpha_equal_to_the_ulp(r["alpha"], params[:, 0])
    _log_flips("quadri / %s / %d-%d-%d" % ("mref" if mode == api.RA_MODE_MREF else "reffree", nx, ou, nref), n, flips)
    eng.close()


def test_quadri_polar_stage_bin_for_bin():
    """Polar2Dm (quadri) -> Normalize_ring -> Frngs of every in-window offset against the oracle, element by element"""
    nx, ou, xr, n = 64, 25, 2, 3
    refs = synth.make_references(2, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 0.5, ou=ou)
    rg = orc.rings(1, ou, 1)
    eng = api.Engine(nx, ou, xr, xr, 1.0, 2, api.RA_MODE_MREF, interp=api.RA_INTERP_QUADRI)
    st = np.zeros((n, 2), np.float32)
    st[1] = (1, -1)
    got = eng.debug_spectra(torch.from_numpy(parts).to(eng.dev), torch.from_numpy(st).to(eng.dev))
    sh = geometry.shift_list(xr, xr, 1.0)
    cnx = nx // 2 + 1
    worst_vs_bilinear = 0.0
    for p in range(n):
        for s in range(len(sh)):
            c = orc.polar2dm(parts[p], cnx + st[p, 0] + sh[s, 0], cnx + st[p, 1] + sh[s, 1], rg, interp=orc.INTERP_QUADRI)
            want = orc.frngs(orc.normalize_ring(c, rg), rg)
            assert np.abs(got[p, s] - want).max() < 1e-5 * np.abs(want).max(), (p, s)
            cb = orc.polar2dm(parts[p], cnx + st[p, 0] + sh[s, 0], cnx + st[p, 1] + sh[s, 1], rg)
            worst_vs_bilinear = max(worst_vs_bilinear, np.abs(orc.frngs(orc.normalize_ring(cb, rg), rg) - want).max() / np.abs(want).max())
    assert worst_vs_bilinear > 1e-3
    eng.close()


def test_options_are_validated():
    with pytest.raises(api.EngineError):
        api.Engine(90, 36, 3, 3, 1.0, 2, interp=7)
    lib = api.load_library()
    import ctypes
    h = ctypes.c_void_p()
    cfg = api.RaConfig(90, 1, 36, 1, 3.0, 3.0, 1.0, 2, api.RA_MODE_MREF, 0, 0)
    opt = api.RaOptions(api.RA_INTERP_BILINEAR, 5)
    assert lib.ra_create_ex(ctypes.byref(h), ctypes.byref(cfg), ctypes.byref(opt)) == -1        # RA_ERR_ARG
    # NULL options = ra_create
    assert lib.ra_create_ex(ctypes.byref(h), ctypes.byref(cfg), None) == 0
    o = api.RaOptions()
    assert lib.ra_get_options(h, ctypes.byref(o)) == 0 and (o.interp, o.normalize_ring) == (0, 1)
    lib.ra_destroy(h)
    # the Python mirror takes None or the header's -1 for "by mode" (bool(-1) must not switch the normalisation on)
    for flag, want in ((None, 0), (-1, 0), (True, 1), (1, 1), (False, 0), (0, 0)):
        eng = api.Engine(90, 36, 3, 3, 1.0, 1, api.RA_MODE_REFFREE, normalize_ring=flag)
        assert eng.options == (api.RA_INTERP_BILINEAR, want), (flag, eng.options)
        eng.set_normalize_ring(True)
        eng.set_normalize_ring(-1)
        assert eng.options[1] == 0
        eng.close()
