"""Regenerates the golden fixtures mref_32.npz / mref_90.npz / reffree_32.npz.

The reference's Python cannot be imported here (EMAN2/sparx absent, Python 2), so the
vectors come from the repository's own CPU restatement (oracle/): inputs are seeded
synthetic stacks, outputs are the oracle's per-particle parameters, peak bins and class sums
for two consecutive iterations.  Run from the repo root: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cryo_ralib_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def update_refs(sums, counts, mask):
    """(even+odd) * (1/n) then normalize.mask(no_sigma=1) (test_mref_gpu_align.py:534-535, 563)"""
    out = []
    for j in range(sums.shape[0]):
        avg = (sums[j, 0] + sums[j, 1]) * np.float32(1.0 / float(counts[j]))
        out.append(orc.normalize_mask(avg, mask, 1))
    return np.stack(out)


def mref_case(nx, ou, nref, n, xr, sigma, name):
    refs = synth.make_references(nref, nx, ou)
    parts, truth = synth.make_particles(refs, n, xr, xr, sigma, ou=ou)
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    refs_n, cref = orc.prepare_refs(refs, mask, rg)
    d = np.zeros((n, 2), np.float32)
    out = dict(particles=parts, refs=refs_n, crefim=cref, ou=ou, xr=xr, truth_cls=truth["cls"], truth_ang=truth["ang"],
               truth_mir=truth["mir"], truth_sx=truth["sx"], truth_sy=truth["sy"])
    cur = refs_n
    for it in range(2):
        _, cref = orc.prepare_refs(cur, None, rg)
        params, infos, sums, counts = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d)
        out["params%d" % it] = params
        out["jtot%d" % it] = np.array([infos[i].jtot for i in range(n)], np.int32)
        out["ixiy%d" % it] = np.array([[infos[i].ix, infos[i].iy] for i in range(n)], np.float32)
        out["state%d" % it] = d.copy()
        out["sums%d" % it] = sums
        out["counts%d" % it] = counts
        if (counts >= 1).all():
            cur = update_refs(sums, counts, mask)
            out["newrefs%d" % it] = cur
    np.savez_compressed(os.path.join(HERE, name), **out)


def reffree_case(nx, ou, n, xr, name):
    refs = synth.make_references(1, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 0.5, ou=ou)
    rg = orc.rings(1, ou, 1)
    tavg = parts.mean(0)[None]
    _, cref = orc.prepare_refs(tavg, None, rg)
    d = np.zeros((n, 2), np.float32)
    params = np.zeros((n, 6), np.float32)
    out = dict(particles=parts, tavg=tavg, ou=ou, xr=xr)
    for it, cs in enumerate([(0.0, 0.0), (0.3, -0.2)]):
        params, infos, sums, ss = orc.reffree_iteration(parts, cref[0], rg, xr, xr, 1.0, cs, d, params)
        out["params%d" % it] = params.copy()
        out["jtot%d" % it] = np.array([infos[i].jtot for i in range(n)], np.int32)
        out["state%d" % it] = d.copy()
        out["sums%d" % it] = sums
        out["cs%d" % it] = np.array(cs, np.float32)
    np.savez_compressed(os.path.join(HERE, name), **out)


if __name__ == "__main__":
    mref_case(32, 12, 3, 8, 2, 0.25, "mref_32.npz")
    mref_case(90, 36, 3, 8, 3, 0.25, "mref_90.npz")
    reffree_case(32, 12, 8, 2, "reffree_32.npz")
    print("golden fixtures written to", HERE)
