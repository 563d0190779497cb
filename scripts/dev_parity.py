"""developer check: HIP engine vs CPU oracle on a small synthetic stack (run on a GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as orc
from cryo_ralib_amd import synth, api

def main(nx=90, ou=36, nref=10, n=64, xr=3, sigma=0.25):
    refs = synth.make_references(nref, nx, ou)
    parts, truth = synth.make_particles(refs, n, xr, xr, sigma, ou=ou)
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    refs_n, cref = orc.prepare_refs(refs, mask, rg)
    d = np.zeros((n, 2), np.float32)
    t = time.time()
    params, infos, sums, counts = orc.mref_iteration(parts, cref, rg, xr, xr, 1, d, nthreads=8)
    print("oracle s/particle", (time.time() - t) / n)

    eng = api.Engine(nx, ou, xr, xr, 1.0, nref, api.RA_MODE_MREF)
    dev = eng.dev
    t_refs = torch.from_numpy(refs_n).to(dev)
    eng.set_references(t_refs)
    gc = eng.prepared_references()
    print("prepared refs max rel err", np.abs(gc - cref).max() / np.abs(cref).max())
    t_parts = torch.from_numpy(parts).to(dev)
    state = eng.new_state(n); res = eng.new_result(n)
    eng.align(t_parts, state, res)
    eng.sync()
    r = eng.result_to_numpy(res)
    st = state.cpu().numpy()
    bad = 0
    for i in range(n):
        ok = (r["ref_id"][i] == int(params[i, 4]) and r["mirror"][i] == int(params[i, 3]) and r["angle_bin"][i] == infos[i].jtot
              and abs(st[i, 0] - d[i, 0]) < 1e-6 and abs(st[i, 1] - d[i, 1]) < 1e-6)
        if not ok:
            bad += 1
            print("MISMATCH", i, r[i], params[i], infos[i].jtot, d[i], st[i])
    rel = np.abs(r["peak"] - params[:, 5]) / np.abs(params[:, 5])
    print("mismatches", bad, "of", n, "peak rel err max", rel.max())
    print("alpha err", np.abs(r["alpha"] - params[:, 0]).max(), "sx err", np.abs(r["sx"] - params[:, 1]).max(), np.abs(r["sy"] - params[:, 2]).max())
    gs = torch.zeros((nref, 2, nx, nx), device=dev); gcnt = torch.zeros(nref, dtype=torch.int32, device=dev)
    al = torch.zeros((n, nx, nx), device=dev)
    eng.transform_accumulate(t_parts, res, 0, al, gs, gcnt)
    eng.sync()
    print("counts equal", (gcnt.cpu().numpy() == counts).all(), "sums max abs err", np.abs(gs.cpu().numpy() - sums).max(), "scale", np.abs(sums).max())
    a0 = orc.rot_shift2d(parts[0], params[0, 0], params[0, 1], params[0, 2], int(params[0, 3]))
    print("aligned[0] err", np.abs(al[0].cpu().numpy() - a0).max())
    newrefs = t_refs.clone()
    eng.update_references(gs, gcnt, newrefs, 1)
    eng.sync()
    print("newrefs finite", torch.isfinite(newrefs).all().item())

if __name__ == "__main__":
    if len(sys.argv) > 1:       # nx ou nref n xr [sigma]
        a = sys.argv[1:]
        main(nx=int(a[0]), ou=int(a[1]), nref=int(a[2]), n=int(a[3]), xr=int(a[4]), sigma=float(a[5]) if len(a) > 5 else 0.25)
    else:
        main()
        main(nx=32, ou=12, nref=3, n=16, xr=2)
