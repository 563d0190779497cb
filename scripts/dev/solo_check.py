"""dev check of search_solo_kernel (maxrin 512): polar stage bin for bin, search against the oracle and against the
generic kernels, a timing.  python scripts/dev/solo_check.py [nx ou nref n sigma]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cryo_ralib_amd import api, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
import test_gpu_parity as T  # noqa: E402


def main():
    nx, ou, nref, n = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (128, 60, 10, 128)))
    sigma = float(sys.argv[5]) if len(sys.argv) > 5 else 0.5
    nbig = int(sys.argv[6]) if len(sys.argv) > 6 else 4096
    xr = 3
    if os.environ.get("SOLO_POLAR", "1") == "1":
        T.polar_stage_check(nx, ou, xr, api.RA_MODE_MREF)
        print("polar stage bin for bin: ok")
    refs = synth.make_references(nref, nx, ou)
    parts, truth = synth.make_particles(refs, n, xr, xr, sigma, ou=ou)
    rg, mask, refs_n, cref = T.oracle_setup(refs, ou, nx)
    d = np.zeros((n, 2), np.float32)
    t0 = time.time()
    params, infos, sums, counts = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d, nthreads=16)
    print("oracle: %.1f s" % (time.time() - t0))
    eng, tp, st, res = T.run_engine(parts, refs_n, ou, xr, xr, 1.0)
    print("search path:", eng.search_path)
    r = eng.result_to_numpy(res)
    flips = T.compare_search(r, st.cpu().numpy(), params, infos, d, max_tie_frac=0.0)
    print("vs oracle: flips", flips, T.LAST_COMPARE)
    # timing on a larger batch (copies of the sample)
    reps = max(1, nbig // n)
    big = tp.repeat(reps, 1, 1).contiguous()
    nb = big.shape[0]
    stb, resb = eng.new_state(nb), eng.new_result(nb)
    for it in range(3):
        stb.zero_()
        torch.cuda.synchronize()
        t0 = time.time()
        eng.align(big, stb, resb)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("align %d particles: %.2f ms = %.0f particles/s" % (nb, dt * 1e3, nb / dt))
    eng.close()


if __name__ == "__main__":
    main()
