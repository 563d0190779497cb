"""developer check (GPU box): which kind of near-tie the engine and the oracle resolve differently on bench.py's parity
samples -- search offset, mirror or angle bin.  python scripts/dev/flip_types.py [reffree|mref] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from cryo_ralib_amd import api, synth
from oracle import oracle as orc

wl = sys.argv[1] if len(sys.argv) > 1 else "reffree"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg, nx, ou, xr, nref, _, _, _ = bench.WORKLOADS[wl]
reffree = wl == "reffree"
dev = torch.device("cuda", 0)
refs_np = synth.make_references(max(nref, 1), nx, ou)
rg = orc.rings(1, ou, 1); mask = orc.model_circle(ou, nx, nx)
for sigma in (0.25, 1.0):
    parts_t, _ = bench.generate_shard(dev, refs_np, n, xr, xr, sigma, 7, nx, ou)
    parts = np.stack([orc.normalize_mask(p, mask, 0) for p in parts_t.cpu().numpy()])
    d = np.zeros((n, 2), np.float32)
    if reffree:
        tavg = parts.mean(0)[None].astype(np.float32)
        refs_n, cref = orc.prepare_refs(tavg, None, rg)
        params, infos, _, _ = orc.reffree_iteration(parts, cref[0], rg, xr, xr, 1.0, (0, 0), d, np.zeros((n, 6), np.float32), nthreads=bench.host_cores())
        mode = api.RA_MODE_REFFREE
    else:
        refs_n, cref = orc.prepare_refs(refs_np, mask, rg)
        params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d, nthreads=bench.host_cores())
        mode = api.RA_MODE_MREF
    eng = api.Engine(nx, ou, xr, xr, 1.0, refs_n.shape[0], mode, device=0)
    eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(dev))
    st, res = eng.new_state(n), eng.new_result(n)
    eng.align(torch.from_numpy(parts).to(dev), st, res)
    eng.sync()
    r = eng.result_to_numpy(res); s = st.cpu().numpy()
    eng.close()
    jt = np.array([infos[i].jtot for i in range(n)])
    off = np.abs(s - d).max(1) >= 1e-6
    ref = r["ref_id"] != params[:, 4].astype(int)
    mir = r["mirror"] != params[:, 3].astype(int)
    ang = r["angle_bin"] != jt
    bad = off | ref | mir | ang
    print("sigma %g: %d flips of %d: offset %d, ref %d, mirror %d, angle bin only %d" % (sigma, bad.sum(), n, off.sum(), (ref & ~off).sum(), (mir & ~off & ~ref).sum(), (ang & ~off & ~ref & ~mir).sum()))
    for i in np.where(bad)[0][:20]:
        print("  p%d gpu: off(%g,%g) ref %d mir %d bin %d peak %.9g | cpu: off(%g,%g) ref %d mir %d bin %d peak %.9g" % (
            i, s[i, 0], s[i, 1], r["ref_id"][i], r["mirror"][i], r["angle_bin"][i], r["peak"][i],
            d[i, 0], d[i, 1], params[i, 4], params[i, 3], jt[i], params[i, 5]))
