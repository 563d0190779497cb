"""Float ties between search OFFSETS at half-pixel steps (found by scripts/dev/random_legacy_sweep.py in round 6).  A ridge of the CCF
in (shift, angle) puts neighbouring offsets within 1e-6 of each other; what orders them in the CPU path is then

  * the float sums of Util::Normalize_ring (av += v w, sq += v v w over lcirc samples: a rounding walk of ~1e-6 of sigma that differs
    from offset to offset) -- exact_candidate repeats it bit for bit (77 / 29: exact maxima 4.7e-7 apart), and
  * the FLOAT `peak` of Util::multiref_polar_ali_2d (peak = static_cast<float>(qn); the next double is compared with the rounded
    value) -- refine_winner_kernel replays the scan over all candidates within the tolerance (108 / 46: both offsets 6459.770508).

Both geometries with the seeded synthetic stacks that showed the disagreement, through the engine API and through the drop-in
symbols, two iterations, the literal bar (tests/test_gpu_parity.py::compare_search)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cryo_ralib_amd import api, synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nx,ou,nref,n", [(77, 29, 5, 30), (108, 46, 4, 25)])
def test_offset_ties_at_half_pixel_steps(nx, ou, nref, n):
    from test_gpu_parity import compare_search
    xr, ts = 1, 0.5
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 0.4, ou=ou)
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    refs_n, cref = orc.prepare_refs(refs, mask, rg)
    eng = api.Engine(nx, ou, xr, xr, ts, nref, api.RA_MODE_MREF)
    eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
    tp = torch.from_numpy(parts).to(eng.dev)
    st, res = eng.new_state(n), eng.new_result(n)
    lib = api.load_library()
    cfg = api.AlignConfig(n, nref, nx, ou, rg.maxrin, ts, float(xr), float(xr))
    prm = ctypes.cast(lib.pre_align_init(n, ctypes.byref(cfg), 0), api.aln_param_ptr)
    lib.pre_align_fetch(api.get_c_ptr_array(list(parts)), n, b"sbj_batch")
    lib.reset_shifts(float(xr), ts)
    d = np.zeros((n, 2), np.float32)
    refined = 0
    for it in range(2):
        params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, ts, d, nthreads=8)
        eng.align(tp, st, res)
        eng.sync()
        refined += eng.last_refine_count()
        assert compare_search(api.Engine.result_to_numpy(res), st.cpu().numpy(), params, infos, d) == 0
        lib.pre_align_fetch(api.get_c_ptr_array(list(refs_n)), nref, b"ref_batch")
        lib.mref_align_run_m(0, n)
        for k in range(n):
            assert prm[k].ref_id == int(params[k, 4]) and prm[k].mirror == bool(params[k, 3]), (it, k)
            assert prm[k].shift_x == d[k, 0] and prm[k].shift_y == d[k, 1], (it, k, prm[k].shift_x, prm[k].shift_y, d[k])
    assert refined >= 1, "the stacks hold at least one offset tie that the exact re-evaluation decides"
    lib.gpu_clear()
    eng.close()
