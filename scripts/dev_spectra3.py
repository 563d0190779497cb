import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as orc
from cryo_ralib_amd import synth, api, geometry
np.set_printoptions(linewidth=200, precision=5, suppress=True)
nx, ou, n, xr = 90, 36, 1, 3
refs = synth.make_references(2, nx, ou)
parts, _ = synth.make_particles(refs, n, xr, xr, 0.5, ou=ou)
rg = orc.rings(1, ou, 1)
eng = api.Engine(nx, ou, xr, xr, 1.0, 1, api.RA_MODE_REFFREE)
st = np.zeros((n, 2), np.float32)
got = eng.debug_spectra(torch.from_numpy(parts).to(eng.dev), torch.from_numpy(st).to(eng.dev))
sh = geometry.shift_list(xr, xr, 1.0)
cnx = nx // 2 + 1
for s in (0, 1, 2, 3, 4):
    c = orc.polar2dm(parts[0], cnx + sh[s, 0], cnx + sh[s, 1], rg)
    print("shift", s, "max err rings>=2 slots", np.abs(got[0, s, 24:] - c[24:])[np.arange(len(c) - 24) % 2 == 0].max())
    print(" want ring0", c[:8]); print(" got  ring0", got[0, s, :8])
    print(" want ring1", c[8:24]); print(" got  ring1", got[0, s, 8:24])
