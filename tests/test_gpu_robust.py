"""Degenerate particles in a batch -- all zeros, a constant, NaN, a single inf, values of 1e30 -- through the kernel families: the call
returns (no fault, no wave that never finishes), and the ordinary particles beside them get bit for bit the records they get without
them.  (What the CPU path returns for such a particle is not defined: its restatement indexes with the NaN-derived bin.)"""
import numpy as np
import pytest
import torch

from cryo_ralib_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nx,ou,xr,nref,mode", [(90, 36, 3, 10, api.RA_MODE_MREF), (90, 36, 3, 1, api.RA_MODE_REFFREE),
                                                (90, 36, 3, 50, api.RA_MODE_MREF), (44, 10, 2, 3, api.RA_MODE_MREF),
                                                (100, 40, 3, 10, api.RA_MODE_MREF), (128, 60, 2, 4, api.RA_MODE_REFFREE),
                                                (150, 66, 2, 3, api.RA_MODE_MREF)])
def test_degenerate_particles_do_not_disturb_their_neighbours(nx, ou, xr, nref, mode):
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
        tp = torch.from_numpy(stack).to(eng.dev)
        eng.align(tp, st, res)
        # rot_shift2D + class sums with the parameters of that search (NaN angles and shifts for the NaN particles: every source
        # position NaN -> the "background" copy of the input pixel, no tap outside the image), with and without the aligned stack
        sums = torch.zeros((max(nref, 1), 2, nx, nx), dtype=torch.float32, device=eng.dev)
        counts = torch.zeros(max(nref, 1), dtype=torch.int32, device=eng.dev)
        aligned = torch.zeros((n, nx, nx), dtype=torch.float32, device=eng.dev)
        eng.transform_accumulate(tp, res, 0, None, sums, counts)
        sums2, counts2 = torch.zeros_like(sums), torch.zeros_like(counts)
        eng.transform_accumulate(tp, res, 0, aligned, sums2, counts2)
        eng.sync()
        out.append((api.Engine.result_to_numpy(res).copy(), st.cpu().numpy().copy(), sums.cpu().numpy(), counts.cpu().numpy(), aligned.cpu().numpy()))
        eng.close()
    keep = [i for i in range(n) if i not in special]
    for f in api.RESULT_DTYPE.names:
        np.testing.assert_array_equal(out[0][0][f][keep], out[1][0][f][keep], err_msg=f)
    np.testing.assert_array_equal(out[0][1][keep], out[1][1][keep])
    # the records of the degenerate particles stay inside the tables (a caller indexes references and offsets with them)
    r = out[1][0]
    assert ((r["ref_id"] >= 0) & (r["ref_id"] < max(nref, 1))).all() and ((r["mirror"] == 0) | (r["mirror"] == 1)).all()
    assert ((r["shift_idx"] >= 0) & (r["shift_idx"] < (2 * xr + 1) ** 2)).all()
    assert np.isfinite(out[1][1]).all()
    assert out[1][3].sum() == n
    # aligned images of the ordinary particles: unchanged; classes without a degenerate member: the same sums
    np.testing.assert_array_equal(out[0][4][keep], out[1][4][keep])
    touched = set(int(r["ref_id"][i]) for i in special) | set(int(out[0][0]["ref_id"][i]) for i in special)
    for c in range(max(nref, 1)):
        if c not in touched:
            np.testing.assert_array_equal(out[0][2][c], out[1][2][c])
