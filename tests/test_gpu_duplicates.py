"""Exact ties between REFERENCES: copies of one reference inside a stack (two, three, four of them; inside one reference tile of the
kernels and across tiles).  Their CCFs are equal to the bit, so the winner is decided by the scan alone -- Util::multiref_polar_ali_2d
walks the references in ascending order and a later one wins with ">=": the LAST copy -- in every kernel family, and after the exact
re-evaluation (refine = -1) as well."""
import numpy as np
import pytest
import torch

from cryo_ralib_amd import api, synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

CASES = [
    # nx, ou, xr, nref, groups of equal references
    (90, 36, 3, 10, [(3, 7, 9)]),                       # search_fused_kernel: one tile
    (90, 36, 3, 50, [(5, 15, 45), (20, 21)]),           # search_tiled_kernel: across and inside tiles
    (44, 10, 2, 4, [(0, 1, 2, 3)]),                     # kernel pair
    (100, 40, 3, 10, [(0, 9), (4, 5, 6)]),              # search_pair_kernel
    (130, 52, 2, 6, [(1, 4)]),                          # search_duo / solo kernels
    (150, 66, 2, 20, [(2, 18), (7, 8, 19)]),            # size-generic class: 16-reference tiles
    (112, 30, 3, 24, [(0, 23), (11, 12)]),              # tiled kernel on a crop
]


@pytest.mark.parametrize("refine", [None, -1.0], ids=["default", "refine-all"])
@pytest.mark.parametrize("nx,ou,xr,nref,groups", CASES, ids=["%d-%d-%d" % (c[0], c[1], c[3]) for c in CASES])
def test_copies_of_a_reference_the_last_one_wins(nx, ou, xr, nref, groups, refine):
    from test_gpu_parity import compare_search
    n = 48
    refs = synth.make_references(nref, nx, ou)
    for g in groups:
        for k in g[1:]:
            refs[k] = refs[g[0]]
    parts, truth = synth.make_particles(refs, n, xr, xr, 0.5, ou=ou)
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    refs_n, cref = orc.prepare_refs(refs, mask, rg)
    d = np.zeros((n, 2), np.float32)
    params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d, nthreads=8)
    eng = api.Engine(nx, ou, xr, xr, 1.0, nref, api.RA_MODE_MREF)
    if refine is not None:
        eng.set_refine(refine)
    eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(eng.dev))
    st, res = eng.new_state(n), eng.new_result(n)
    eng.align(torch.from_numpy(parts).to(eng.dev), st, res)
    eng.sync()
    r = api.Engine.result_to_numpy(res)
    assert compare_search(r, st.cpu().numpy(), params, infos, d) == 0
    # the stack does hold particles of the copied classes, and they went to the last copy
    last = {k: g[-1] for g in groups for k in g}
    hit = [i for i in range(n) if int(truth["cls"][i]) in last]
    assert len(hit) >= 3
    assert all(int(r["ref_id"][i]) == last[int(truth["cls"][i])] for i in hit if int(params[i, 4]) in last.values())
    eng.close()
