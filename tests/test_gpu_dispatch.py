"""The dispatch boundaries of ra_create, pinned: ra_align picks among seven kernel families by the LDS plan of the geometry
(include/ralign.h: ra_search_path), and a plan that is off by one ring, one reference or one pixel column lands in another family.
A FIXED, seeded list -- one case on each side of every boundary (outer radius 36 | 37, 39 | 40, 40 | 41, 60 | 61; references
14 | 15 and 16 | 17; boxes 93 | 94 and 140 | 141; half-pixel steps, inner radius 3, ring skip 2, both modes) -- asserts WHICH family
runs and, against the CPU oracle, zero disagreements of the integer assignments, CCF peaks within 1e-4 and the class sums of
rot_shift2D (whose own kernels change at ~100 and ~140 pixels).  (the random sweeps of scripts/dev/random_sweep.py
stay a development tool.)

`python tests/test_gpu_dispatch.py` on a GPU box prints the family every case takes (how the table below was filled).
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))       # (run as a script: see the last lines)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cryo_ralib_amd import api, synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

M, F = api.RA_MODE_MREF, api.RA_MODE_REFFREE
# search families: (ra_search_path, ra_search_tiled, ra_search_offsets_per_pass)
FUSED, TILED, PAIRK, SOLO2, PAIR2, GENERIC = (1, 0, 0), (1, 1, 0), (0, 0, 0), (3, 0, 2), (3, 0, 2), (2, 0, 0)

CASES = [
    # name,                       nx,  ou, ir, rs, xr, yr, ts,  nref, n, mode, family
    ("box93-ou36",                93,  36, 1, 1, 3, 3, 1.0, 10, 16, M, FUSED),       # whole image next to four ring buffers
    ("box94-ou36",                94,  36, 1, 1, 3, 3, 1.0, 10, 16, M, FUSED),       # ... a crop of it (tcrop)
    ("box128-ou36",               128, 36, 1, 1, 3, 3, 1.0, 10, 12, M, FUSED),
    ("box128-ou37-tight",         128, 37, 1, 1, 3, 3, 1.0, 10, 12, M, FUSED),       # rings 4 floats apart
    ("box128-ou37-R20",           128, 37, 1, 1, 3, 3, 1.0, 20, 8, M, PAIR2),        # ... which the tiled plan does not fit: pair kernel
    ("box128-ou39-R8",            128, 39, 1, 1, 3, 3, 1.0, 8, 12, M, FUSED),
    ("box128-ou40-R8",            128, 40, 1, 1, 3, 3, 1.0, 8, 12, M, PAIR2),        # ou = 40: the one 256-sample radius without four buffers
    ("box100-ou40-reffree",       100, 40, 1, 1, 3, 3, 1.0, 1, 16, F, PAIR2),
    ("box100-ou41",               100, 41, 1, 1, 2, 2, 1.0, 4, 10, M, SOLO2),        # first radius with 512-sample rings: duo kernel
    ("box140-ou60",               140, 60, 1, 1, 2, 2, 1.0, 3, 6, M, SOLO2),         # last radius whose image crop + one ring buffer fit the LDS
    ("box140-ou61",               140, 61, 1, 1, 2, 2, 1.0, 3, 6, M, GENERIC),
    ("box160-ou70",               160, 70, 1, 1, 2, 2, 1.0, 2, 4, M, GENERIC),       # more than 64 rings
    ("R14",                       90,  36, 1, 1, 3, 3, 1.0, 14, 16, M, FUSED),
    ("R15",                       90,  36, 1, 1, 3, 3, 1.0, 15, 16, M, TILED),
    ("R16-maxrin128",             48,  20, 1, 1, 2, 2, 1.0, 16, 16, M, FUSED),       # no tiled kernel at maxrin 128: fused up to 16,
    ("R17-maxrin128",             48,  20, 1, 1, 2, 2, 1.0, 17, 16, M, PAIRK),       # ... the kernel pair beyond
    ("R16-crop",                  128, 36, 1, 1, 3, 3, 1.0, 16, 8, M, TILED),
    ("R17-crop",                  128, 36, 1, 1, 3, 3, 1.0, 17, 8, M, TILED),
    ("box128-ou25-R50",           128, 25, 1, 1, 3, 3, 1.0, 50, 8, M, TILED),        # image fits the LDS, the tiled plan beside it does not: tiled kernel
    ("box100-ou30-R17",           100, 30, 1, 1, 3, 3, 1.0, 17, 8, M, TILED),        # ... over a crop (planned in the size-generic class; was the kernel pair)
    ("half-pixel-steps",          90,  36, 1, 1, 1, 1, 0.5, 5, 16, M, FUSED),
    ("half-pixel-steps-crop",     112, 34, 1, 1, 1, 2, 0.5, 3, 12, M, FUSED),
    ("half-pixel-steps-duo",      120, 50, 1, 1, 1, 1, 0.5, 2, 6, M, SOLO2),
    ("inner-radius-3",            90,  36, 3, 1, 3, 3, 1.0, 6, 16, M, FUSED),
    ("inner-radius-3-pair",       104, 40, 3, 1, 2, 3, 1.0, 3, 10, M, PAIR2),
    ("ring-skip-2",               90,  36, 1, 2, 3, 2, 1.0, 4, 16, M, FUSED),
    ("ring-skip-2-reffree",       76,  30, 2, 2, 2, 2, 1.0, 1, 16, F, FUSED),
    ("ring-skip-2-big",           150, 70, 1, 2, 2, 2, 1.0, 2, 4, M, SOLO2),         # 35 rings of up to 512 samples: the duo kernel at radius 70
    ("reffree-90-36",             90,  36, 1, 1, 3, 3, 1.0, 1, 24, F, FUSED),
    ("reffree-crop-128-36",       128, 36, 1, 1, 3, 3, 1.0, 1, 16, F, FUSED),
    ("reffree-duo-128-60",        128, 60, 1, 1, 2, 2, 1.0, 1, 8, F, SOLO2),
    ("tiny-rings",                40,  9,  3, 2, 3, 1, 1.0, 1, 12, F, PAIRK),        # rings below 8 samples' worth of the fused jobs
    ("generic-ir3-rs2",           176, 80, 3, 2, 2, 2, 1.0, 2, 4, M, GENERIC),       # ring zones over every second ring from radius 3 (polar_zone_kernel)
    ("generic-half-pixel",        150, 66, 1, 1, 1, 1, 0.5, 2, 4, M, GENERIC),       # ... fractional sampling centres
    ("generic-reffree",           160, 70, 1, 1, 2, 2, 1.0, 1, 4, F, GENERIC),       # ... without Normalize_ring
    ("box140-sums",               140, 36, 1, 1, 3, 3, 1.0, 3, 8, M, FUSED),         # transform_sum_kernel in row bands
    ("box141-sums",               141, 36, 1, 1, 3, 3, 1.0, 3, 8, M, FUSED),         # transform_sum_tile_kernel
]
IDS = [c[0] for c in CASES]


def _run(case, check=True):
    name, nx, ou, ir, rs, xr, yr, ts, nref, n, mode, family = case
    refs = synth.make_references(nref, nx, ou, seed=1000 + IDS.index(name))
    parts, _ = synth.make_particles(refs, n, xr, yr, 0.5, shard=IDS.index(name), ou=ou)          # seeded: 2000 + shard
    eng = api.Engine(nx, ou, xr, yr, ts, nref, mode, first_ring=ir, ring_skip=rs)
    got_family = (eng.search_path, int(eng.search_tiled), eng.search_offsets_per_pass)
    if not check:
        eng.close()
        return got_family
    rg = orc.rings(ir, ou, rs)
    mask = orc.model_circle(ou, nx, nx)
    refs_n, cref = orc.prepare_refs(refs, mask, rg)
    d = np.zeros((n, 2), np.float32)
    if mode == M:
        params, infos, sums, counts = orc.mref_iteration(parts, cref, rg, xr, yr, ts, d, nthreads=8)
    else:
        p0 = np.zeros((n, 6), np.float32)
        params, infos, sums, _ = orc.reffree_iteration(parts, cref[0], rg, xr, yr, ts, (0, 0), d, p0, nthreads=8)
        counts = np.array([n], np.int32)          # one class: every particle
    eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
    tp = torch.from_numpy(parts).to(eng.dev)
    st, res = eng.new_state(n), eng.new_result(n)
    eng.align(tp, st, res)
    gs = torch.zeros((nref, 2, nx, nx), device=eng.dev)
    gc = torch.zeros(nref, dtype=torch.int32, device=eng.dev)
    eng.transform_accumulate(tp, res, 0, None, gs, gc)
    eng.sync()
    r = api.Engine.result_to_numpy(res)
    out = (got_family, r, st.cpu().numpy(), params, infos, d, gs.cpu().numpy(), gc.cpu().numpy(), sums, counts, mask)
    eng.close()
    return out


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_dispatch_boundary(case):
    from test_gpu_parity import compare_search, default_path_only, _log_flips
    default_path_only("RALIGN_FUSED", "RALIGN_TILED", "RALIGN_GENERIC", "RALIGN_PAIR", "RALIGN_SOLO", "RALIGN_DUO", "RALIGN_TCROP",
                      "RALIGN_CROP", "RALIGN_TIGHT_RINGS")
    got_family, r, st, params, infos, d, gs, gc, sums, counts, mask = _run(case)
    assert got_family == case[-1], "%s: kernel family %s, expected %s" % (case[0], got_family, case[-1])
    flips = compare_search(r, st, params, infos, d)
    assert flips == 0
    _log_flips("dispatch boundary " + case[0], case[9], flips)
    assert (gc == np.asarray(counts).reshape(-1)).all()
    # class sums of a handful of particles: rot_shift2D's quadratic interpolant is not continuous across pixel-cell borders, so the
    # 1e-5 degree between an f32 and an f64 sub-bin angle moves single pixels by O(sigma) (assert_images_close of
    # tests/test_gpu_parity.py, which asks 2e-4 of sums over hundreds of particles); with 1 - 3 particles per sum the bar is 99.5 % of
    # the pixels within 2e-3 and 3e-3 overall -- a wrong kernel at a box-size boundary is off by O(1)
    want = np.asarray(sums).reshape(gs.shape)
    diff = np.abs(gs - want)
    sel = np.broadcast_to(mask > 0.5, diff.shape)
    assert np.quantile(diff[sel], 0.995) < 2e-3, np.quantile(diff[sel], 0.995)
    assert np.linalg.norm(diff[sel]) < 3e-3 * np.linalg.norm(want[sel]) + 1e-6, np.linalg.norm(diff[sel]) / np.linalg.norm(want[sel])


def test_reset_shifts_replans_the_ring_zones():
    """the annuli of polar_zone_kernel follow the search range: an engine of the size-generic class created for xr = 3 searches a
    window of xr = 1 after reset_shifts (zones planned again for the narrower range) and agrees with the oracle there"""
    from test_gpu_parity import compare_search, default_path_only
    default_path_only("RALIGN_GENERIC", "RALIGN_SOLO", "RALIGN_ZONES")
    nx, ou, nref, n = 150, 66, 2, 5
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, 1, 1, 0.5, ou=ou)
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    refs_n, cref = orc.prepare_refs(refs, mask, rg)
    eng = api.Engine(nx, ou, 3, 3, 1.0, nref, M)
    assert eng.search_path == 2
    eng.reset_shifts(1, 1, 1.0)
    assert eng.num_shifts == 9
    d = np.zeros((n, 2), np.float32)
    params, infos, _, _ = orc.mref_iteration(parts, cref, rg, 1, 1, 1.0, d, nthreads=8)
    eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
    st, res = eng.new_state(n), eng.new_result(n)
    eng.align(torch.from_numpy(parts).to(eng.dev), st, res)
    eng.sync()
    assert compare_search(api.Engine.result_to_numpy(res), st.cpu().numpy(), params, infos, d) == 0
    eng.close()


@pytest.mark.parametrize("mode", [M, F], ids=["mref", "reffree"])
def test_live_offset_lists_of_the_generic_class(mode, monkeypatch):
    """the size-generic kernels work on the in-window offsets of every particle only (live_scan_kernel, DevGeom::ent_base): particles
    whose state sits at the edge of the allowed shifts -- windows of 2 .. 7 offsets per axis instead of 7, one reset by the mashi rule --
    against the oracle, and bitwise against the engine that computes every offset and masks afterwards (RALIGN_LIVE_OFFSETS=0)"""
    from test_gpu_parity import compare_search, default_path_only
    default_path_only("RALIGN_GENERIC", "RALIGN_SOLO", "RALIGN_LIVE_OFFSETS")
    nx, ou, xr, n = 150, 66, 3, 9
    nref = 3 if mode == M else 1
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, 1, 1, 0.5, ou=ou)
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    refs_n, cref = orc.prepare_refs(refs, mask if mode == M else None, rg)
    d0 = np.array([[0, 0], [7, 0], [-7, 7], [8, -8], [0, -6], [5, 8], [-8, 3], [9, 0], [-3, -7]], np.float32)      # mashi = 8; (9, 0): reset / clamped
    d = d0.copy()
    if mode == M:
        params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d, nthreads=8)
    else:
        # (ali2d_single_iter takes its starting shift from the previous parameters: inverse_transform2(0, -dx, -dy) = (dx, dy))
        p0 = np.zeros((n, 6), np.float32)
        p0[:, 1:3] = -d0
        params, infos, _, _ = orc.reffree_iteration(parts, cref[0], rg, xr, xr, 1.0, (0, 0), d, p0, nthreads=8)
    out = {}
    for live in ("1", "0"):
        monkeypatch.setenv("RALIGN_LIVE_OFFSETS", live)
        eng = api.Engine(nx, ou, xr, xr, 1.0, nref, mode)
        assert eng.search_path == 2 and eng.search_skips_offsets == (live == "1")
        eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
        st = torch.from_numpy(d0.copy()).to(eng.dev)
        res = eng.new_result(n)
        eng.align(torch.from_numpy(parts).to(eng.dev), st, res)
        eng.sync()
        out[live] = (api.Engine.result_to_numpy(res), st.cpu().numpy())
        assert compare_search(out[live][0], out[live][1], params, infos, d) == 0
        # the polar stage of the in-window offsets, bin for bin (out-of-window offsets: zeros with live lists)
        eng.close()
    for k in ("ref_id", "mirror", "angle_bin", "shift_idx", "peak", "alpha", "sx", "sy"):
        np.testing.assert_array_equal(out["1"][0][k], out["0"][0][k])
    np.testing.assert_array_equal(out["1"][1], out["0"][1])


@pytest.mark.parametrize("mode", [M, F])
@pytest.mark.parametrize("nx,ou,ir,rs", [(73, 15, 2, 2), (90, 36, 1, 3), (150, 66, 2, 3)])
def test_window_radius_with_a_ring_step(mode, nx, ou, ir, rs):
    """a ring step that does not divide last_ring - first_ring leaves the outermost sampled ring below last_ring: mref_ali2d resets
    and cuts its windows with the last_ring ARGUMENT (test_mref_gpu_align.py:740, 761-766), ali2d_single_iter clamps with
    ou = numr[-3] (DevGeom::win_ring).  States at mashi, between the two mashis and beyond, against the oracle"""
    from test_gpu_parity import compare_search
    rg = orc.rings(ir, ou, rs)
    act = rg.numr[3 * (rg.nring - 1)]
    assert act < ou
    xr, n = 2, 10
    nref = 3 if mode == M else 1
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, 1, 1, 0.3, ou=ou)
    cn = nx // 2 + 1
    m_arg, m_act = cn - ou - 2, cn - act - 2
    d0 = np.array([[m_arg, 0], [m_arg + 0.5, 0], [0, -(m_arg + 1)], [m_act, m_act], [-(m_act + 0.5), 0], [0, m_act + 1],
                   [-m_arg, m_arg - 1], [m_arg - 1.5, -m_act], [0, 0], [m_act + 3, -1]], np.float32)
    for i in range(n):      # the particle sits where its state says
        parts[i] = np.roll(parts[i], (int(np.floor(d0[i, 1])), int(np.floor(d0[i, 0]))), axis=(0, 1))
    mask = orc.model_circle(ou, nx, nx)
    refs_n, cref = orc.prepare_refs(refs, mask if mode == M else None, rg)
    d = d0.copy()
    if mode == M:
        params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d, nthreads=8)
    else:
        p0 = np.zeros((n, 6), np.float32)
        p0[:, 1:3] = -d0
        params, infos, _, _ = orc.reffree_iteration(parts, cref[0], rg, xr, xr, 1.0, (0, 0), d, p0, nthreads=8)
    eng = api.Engine(nx, ou, xr, xr, 1.0, nref, mode, first_ring=ir, ring_skip=rs)
    eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
    st = torch.from_numpy(d0.copy()).to(eng.dev)
    res = eng.new_result(n)
    eng.align(torch.from_numpy(parts).to(eng.dev), st, res)
    eng.sync()
    assert compare_search(api.Engine.result_to_numpy(res), st.cpu().numpy(), params, infos, d) == 0
    eng.close()


def test_generic_class_in_the_iteration_loop():
    """three iterations of the host driver (search, class sums, reference update, state round trip) at a geometry of the size-generic
    class -- ring zones, live-offset lists whose windows move with the states from the second iteration on, one refine launch for the
    whole call over several chunks -- against the same loop built from oracle calls (every particle refined: alpha to the ulp)"""
    from test_gpu_parity import compare_search, default_path_only, assert_alpha_equal_to_the_ulp, _oracle_mref_loop_step, _log_flips
    from cryo_ralib_amd.mref import MrefAligner
    default_path_only("RALIGN_GENERIC", "RALIGN_SOLO", "RALIGN_LIVE_OFFSETS", "RALIGN_ZONES")
    nx, ou, nref, xr, n = 150, 68, 3, 3, 40
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 0.5, ou=ou)
    al = MrefAligner(parts, refs, ou, xr, xr, 1.0, preprocess=True, state_roundtrip=True, refine=-1, chunk=16)       # 3 chunks per call
    assert al.engine.search_path == 2 and al.engine.search_skips_offsets
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
        _log_flips("generic-class loop it=%d" % it, n, flips)
        assert flips == 0
        prev = params
        np.testing.assert_array_equal(got_counts, counts)
        assert_alpha_equal_to_the_ulp(r["alpha"], params[:, 0])
        cur = np.stack([orc.normalize_mask((sums[j, 0] + sums[j, 1]) * np.float32(1.0 / float(counts[j])), mask, 1) for j in range(nref)])
        np.testing.assert_allclose(al.refs.cpu().numpy(), cur, rtol=0, atol=3e-6 * np.abs(cur).max())
    # the windows did move: some particle's state is off centre by now (mashi = 76 - 68 - 2 = 6 > xr: windows shrink from |d| = 4 on)
    assert np.abs(al.state.cpu().numpy()).max() >= 1
    al.close()


if __name__ == "__main__":
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    for c in CASES:
        fam = _run(c, check=False)
        print("%-26s family %s %s" % (c[0], fam, "" if fam == c[-1] else "   <-- table says %s" % (c[-1],)), flush=True)
