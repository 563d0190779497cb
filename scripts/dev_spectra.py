"""developer check: polar / ring-FFT stage of the engine vs Polar2Dm -> Normalize_ring -> Frngs of the oracle"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as orc
from cryo_ralib_amd import synth, api, geometry

def main(nx=90, ou=36, n=3, xr=3):
    refs = synth.make_references(2, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 0.5, ou=ou)
    rg = orc.rings(1, ou, 1)
    eng = api.Engine(nx, ou, xr, xr, 1.0, 2)
    st = np.zeros((n, 2), np.float32); st[1] = (2, -1)
    got = eng.debug_spectra(torch.from_numpy(parts).to(eng.dev), torch.from_numpy(st).to(eng.dev))
    sh = geometry.shift_list(xr, xr, 1.0)
    cnx = nx // 2 + 1
    numr = rg.numr_list()
    worst = 0
    for p in range(n):
        for s in range(len(sh)):
            c = orc.polar2dm(parts[p], cnx + st[p, 0] + sh[s, 0], cnx + st[p, 1] + sh[s, 1], rg)
            want = orc.frngs(orc.normalize_ring(c, rg), rg)
            err = np.abs(got[p, s] - want)
            rel = err.max() / np.abs(want).max()
            if rel > 1e-5:
                ring = [i for i in range(rg.nring) if numr[3*i+1]-1 <= err.argmax() < numr[3*i+1]-1+numr[3*i+2]][0]
                print("particle", p, "shift", s, "rel", rel, "worst idx", err.argmax(), "ring", ring, "len", numr[3*ring+2], "slot", err.argmax() - (numr[3*ring+1]-1))
            worst = max(worst, rel)
    print("worst rel err", worst)

if __name__ == "__main__":
    main()
    main(32, 12, 2, 2)
