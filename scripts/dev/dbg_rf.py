import numpy as np, torch, sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from cryo_ralib_amd import api, synth
from cryo_ralib_amd.mref import RefFreeAligner
from oracle import oracle as orc
import test_gpu_parity as T
nx, ou, xr, n = 90, 36, 3, 256
refs = synth.make_references(1, nx, ou)
parts, _ = synth.make_particles(refs, n, xr, xr, 0.5, ou=ou)
rg = orc.rings(1, ou, 1); mask = orc.model_circle(ou, nx, nx)
al = RefFreeAligner(parts, ou, xr, xr, 1.0)
sums = np.zeros((1, 2, nx, nx), np.float32)
for i in range(n): sums[0, i % 2] += parts[i]
ss = np.zeros(2); params = np.zeros((n, 6), np.float32); d = np.zeros((n, 2), np.float32)
for it in range(3):
    want_tavg, want_a1, want_cs, _, _ = T._oracle_reffree_average(sums, n, mask, ss, it, -1, None)
    a1 = al.iterate(-1, None); al.engine.sync()
    _, cref = orc.prepare_refs(want_tavg[None], None, rg)
    osums = np.zeros((1, 2, nx, nx), np.float32)
    params, infos, osums, oss = orc.reffree_iteration(parts, cref[0], rg, xr, xr, 1.0, want_cs, d, params, sums=osums, nthreads=16)
    sums, ss = osums, np.array(oss)
    r = al.params(); st = al.state.cpu().numpy()
    jt = np.array([infos[i].jtot for i in range(n)])
    print(it, 'cs', al.cs, want_cs, 'mirror eq', (r["mirror"] == params[:, 3].astype(int)).mean(), 'bin eq', (r["angle_bin"] == jt).mean(), 'd maxdiff', np.abs(st - d).max(), 'd eq', (np.abs(st-d).max(1)<1e-6).mean())
    k = np.argmax(np.abs(st-d).max(1)); print('  worst', k, st[k], d[k], r[k], params[k])
