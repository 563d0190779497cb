"""random geometries through the class-resident search (ra_set_class_references + ra_align_classes: one launch, a reference per
particle; ISAC's surface) against the same engine run class by class (ra_set_references + ra_align, the path the other sweeps hold
against the checker) -- bit for bit -- and one class of every case against the checker itself.
python scripts/dev/random_classes_sweep.py [ncase] [seed] [small|big|huge]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cryo_ralib_amd import api, synth          # noqa: E402
from oracle import oracle as orc               # noqa: E402
from test_gpu_parity import compare_search     # noqa: E402

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
size = sys.argv[3] if len(sys.argv) > 3 else "small"
dev = torch.device("cuda:0")
for case in range(ncase):
    xr = int(rng.integers(0, 4)); yr = int(rng.integers(0, 4))
    nx = int(rng.integers(140, 200)) if size == "huge" else int(rng.integers(64, 161)) if size == "big" else int(rng.integers(32, 101))
    oumax = (nx - 1) // 2 - max(xr, yr) - 1
    ou = int(rng.integers(61, min(90, oumax) + 1)) if size == "huge" else int(rng.integers(24, min(78, oumax) + 1)) if size == "big" \
        else int(rng.integers(8, min(40, oumax) + 1))
    ir = int(rng.integers(1, 3)); rs = int(rng.integers(1, 3))
    ts = float(rng.choice([1.0, 1.0, 0.5]))
    ncls = int(rng.integers(2, 5)) if size == "huge" else int(rng.integers(2, 9))
    sizes = [int(rng.integers(1, 5)) if size == "huge" else int(rng.integers(1, 40)) for _ in range(ncls)]
    nomirror = bool(rng.random() < 0.25)
    print("case %2d: nx=%d ou=%d ir=%d rs=%d xr=%d yr=%d ts=%g classes %s nomirror=%d" % (case, nx, ou, ir, rs, xr, yr, ts, sizes, nomirror), flush=True)
    refs = synth.make_references(ncls, nx, ou)
    parts, cls = [], []
    for c, m in enumerate(sizes):
        p, _ = synth.make_particles(refs[c:c + 1], m, max(xr, 1), max(yr, 1), 0.5, shard=c, ou=ou)
        parts.append(p); cls += [c] * m
    parts = np.concatenate(parts); n = len(cls)
    # a random start state inside the box (multiples of the step)
    mashi = nx // 2 + 1 - ou - 2
    lim = int(min(mashi, 3) / ts)
    d0 = (rng.integers(-lim, lim + 1, size=(n, 2)) * ts).astype(np.float32)
    eng = api.Engine(nx, ou, xr, yr, ts, 1, api.RA_MODE_REFFREE, first_ring=ir, ring_skip=rs)
    if nomirror:
        eng.set_nomirror(True)
    tp = torch.from_numpy(parts).to(dev); tr = torch.from_numpy(refs).to(dev)
    tc = torch.tensor(cls, dtype=torch.int32, device=dev)
    st1, res1 = torch.from_numpy(d0.copy()).to(dev), eng.new_result(n)
    try:
        eng.set_class_references(tr)
    except api.EngineError as exc:          # documented: RA_ERR_STATE outside the fused kernel's class, callers loop over the classes
        print("   path %d: no class-resident launch for this geometry (%s)" % (eng.search_path, str(exc)[:60]), flush=True)
        eng.close()
        continue
    eng.align_classes(tp, st1, res1, tc)
    eng.sync()
    st2, res2 = torch.from_numpy(d0.copy()).to(dev), eng.new_result(n)
    start = 0
    for c, m in enumerate(sizes):
        eng.set_references(tr[c:c + 1])
        eng.align(tp[start:start + m], st2[start:start + m], res2[start:start + m])
        start += m
    eng.sync()
    assert torch.equal(st1, st2), "states differ"
    r1, r2 = api.Engine.result_to_numpy(res1), api.Engine.result_to_numpy(res2)
    for f in api.RESULT_DTYPE.names:
        np.testing.assert_array_equal(r1[f], r2[f], err_msg=f)
    # the largest class against the checker
    c = int(np.argmax(sizes)); lo = int(np.sum(sizes[:c])); m = sizes[c]
    rg = orc.rings(ir, ou, rs)
    _, cref = orc.prepare_refs(refs[c:c + 1], None, rg)
    d = d0[lo:lo + m].copy()
    p0 = np.zeros((m, 6), np.float32); p0[:, 1:3] = -d
    orc.set_nomirror(nomirror)
    try:
        params, infos, _, _ = orc.reffree_iteration(parts[lo:lo + m], cref[0], rg, xr, yr, ts, (0, 0), d, p0, nthreads=16)
    finally:
        orc.set_nomirror(False)
    assert compare_search(r1[lo:lo + m], st1.cpu().numpy()[lo:lo + m], params, infos, d) == 0
    print("   path %d (%d offsets per pass)" % (eng.search_path, eng.search_offsets_per_pass), flush=True)
    eng.close()
print("all %d cases agree" % ncase)
