"""Large parity audit: per workload >= 65 536 synthetic particles at sigma = 1.0 through the engine and through
the oracle (all host cores); every disagreement of the integer assignment (ref, mirror, angle bin, shift) is counted and classified:

  neighbour_bin     same reference / mirror / offset, angle bins 1 .. 3 apart (a flat peak: the class finalize_kernel hands to the
                    exact re-evaluation through the 7-point record)
  separated_bins    same reference / mirror / offset, bins further apart (two separated maxima of ONE CCF within 3e-6: not detected
                    by the kernels -- DESIGN.md section 2)
  mirror            same reference and offset, the other mirror half
  other_reference   same offset, another reference (inside one reference tile of the tiled kernel: not detected there)
  other_offset      another search offset

    python scripts/parity_audit_large.py [--n 65536] [--out gpurun_out/parity_audit_large.json] [workloads...]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cryo_ralib_amd import api, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

CASES = {
    # name: (nx, ou, xr, nref, reference-free variant or None)
    "mref": (90, 36, 3, 10, None),                      # BASELINE configs[1]
    "reffree_iter0": (90, 36, 3, 1, "blob"),            # configs[2], first iteration: the reference is the mean of the raw stack
    "reffree_aligned": (90, 36, 3, 1, "template"),      # configs[2], a later iteration: the reference is a sharp average
    "mref50": (90, 36, 3, 50, None),                    # configs[3]
    "nb00": (130, 52, 3, 50, None),                     # the reference notebook's geometry (search_solo_kernel)
    "box128": (128, 60, 3, 10, None),
    "box100": (100, 40, 3, 10, None),                   # search_pair_kernel
    "box96": (96, 36, 3, 10, None),                     # search_pair_kernel with ring buffers grown to hold a tile's spectra
    "box128r38": (128, 38, 3, 10, None),                # search_fused_kernel on a crop with the rings 4 floats apart
    "box256": (256, 36, 3, 10, None),                   # search_pair_kernel on a crop of the image around the particle's centre
    # half-pixel steps: a ridge of the CCF in (shift, angle) puts neighbouring offsets within 1e-6 of each other far more often than
    # whole-pixel steps do (round 6: the float sums of Normalize_ring and the float `peak` of multiref_polar_ali_2d decide those)
    "mref_half": (90, 36, 2, 10, None, 0.5),            # search_fused_kernel, 81 offsets
    "mref_half_xr1": (90, 36, 1, 10, None, 0.5),        # 25 offsets
    "reffree_half": (90, 36, 2, 1, "template", 0.5),
    "box100_half": (100, 40, 1, 10, None, 0.5),         # search_pair_kernel
    "nb00_half": (130, 52, 1, 20, None, 0.5),           # search_solo / duo kernels
    "mref_quarter": (90, 36, 1, 10, None, 0.25),        # quarter-pixel steps, 81 offsets: the ridge holds more near-ties per particle
    "mref_eighth": (90, 36, 0.5, 10, None, 0.125),      # eighth-pixel steps, 81 offsets (stress of the candidate list: RA_TIE_ALTS)
    "reffree_quarter": (90, 36, 1, 1, "template", 0.25),
    "box150_half": (150, 66, 1, 10, None, 0.5),         # size-generic kernels (ring zones), lcirc = 24 k samples
    "largebox_half": (256, 120, 1, 20, None, 0.5),      # configs[4] geometry at half-pixel steps, lcirc = 71 k samples
    "largebox": (256, 120, 5, 100, None),               # BASELINE configs[4] geometry (generic kernels); at most 8192 particles: the oracle needs ~0.5 s of 16 threads each
}
CAPS = {"largebox": 8192, "box256": 16384, "largebox_half": 8192, "box150_half": 32768}


def classify(r, params, jt, d_new, d_old, shifts, maxrin):
    same = (r["ref_id"] == params[:, 4].astype(int)) & (r["mirror"] == params[:, 3].astype(int)) & (r["angle_bin"] == jt)
    osh = d_new - d_old                                   # the oracle's winning offset
    gsh = shifts[r["shift_idx"]]
    same_off = np.abs(osh - gsh).max(1) < 1e-6
    same &= same_off
    out = {"neighbour_bin": 0, "separated_bins": 0, "mirror": 0, "other_reference": 0, "other_offset": 0}
    for i in np.where(~same)[0]:
        if not same_off[i]:
            out["other_offset"] += 1
        elif r["ref_id"][i] != int(params[i, 4]):
            out["other_reference"] += 1
        elif r["mirror"][i] != int(params[i, 3]):
            out["mirror"] += 1
        else:
            db = abs(int(r["angle_bin"][i]) - int(jt[i]))
            db = min(db, maxrin - db)
            out["neighbour_bin" if db <= 3 else "separated_bins"] += 1
    return int((~same).sum()), out


def run_case(name, n, sigma, dev, threads, interp=0, normalize=None):
    """interp / normalize: the engine options of include/ralign.h (ra_options) against the oracle's own switches -- 1 = Util::quadri in
    alrl_ms, normalize = False / True = Normalize_ring off / on whatever the mode (None: the mode's default)"""
    from cryo_ralib_amd import geometry
    nx, ou, xr, nref, rf = CASES[name][:5]
    ts = CASES[name][5] if len(CASES[name]) > 5 else 1.0
    refs_np = synth.make_references(max(nref, 1), nx, ou)
    parts_t, _ = bench.generate_shard(dev, refs_np, n, xr, xr, sigma, 11, nx, ou)
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    parts = parts_t.cpu().numpy()
    del parts_t
    parts = np.stack([orc.normalize_mask(p, mask, 0) for p in parts])
    d = np.zeros((n, 2), np.float32)
    t0 = time.time()
    if rf:
        tavg = parts.mean(0)[None].astype(np.float32) if rf == "blob" else refs_np[:1]
        refs_n, cref = orc.prepare_refs(tavg, None, rg, interp=interp)
        # (normalize: ormq on normalised rings -- the checker's switch; the one-reference multi-reference search is NOT the same thing
        # at float ties: Util::multiref_polar_ali_2d keeps its running peak as a float, ormq as a double)
        orc.set_ormq_normalize(bool(normalize))
        try:
            params, infos, _, _ = orc.reffree_iteration(parts, cref[0], rg, xr, xr, ts, (0, 0), d, np.zeros((n, 6), np.float32), nthreads=threads, interp=interp)
        finally:
            orc.set_ormq_normalize(False)
        mode = api.RA_MODE_REFFREE
    else:
        refs_n, cref = orc.prepare_refs(refs_np, mask, rg, interp=interp)
        params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, ts, d, nthreads=threads, interp=interp, normalize=True if normalize is None else bool(normalize))
        mode = api.RA_MODE_MREF
    t_or = time.time() - t0
    eng = api.Engine(nx, ou, xr, xr, ts, refs_n.shape[0], mode, device=dev.index, interp=interp, normalize_ring=normalize)
    eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(dev))
    st, res = eng.new_state(n), eng.new_result(n)
    eng.align(torch.from_numpy(parts).to(dev), st, res)
    eng.sync()
    refined = eng.last_refine_count() if hasattr(eng, "last_refine_count") else None
    r = eng.result_to_numpy(res)
    path, tiled, maxrin = eng.search_path, eng.search_tiled, eng.maxrin
    eng.close()
    jt = np.array([infos[i].jtot for i in range(n)])
    shifts = geometry.shift_list(xr, xr, ts).astype(np.float32)
    flips, kinds = classify(r, params, jt, d, np.zeros_like(d), shifts, maxrin)
    rel = np.abs(r["peak"] - params[:, 5]) / np.abs(params[:, 5])
    ok = (r["ref_id"] == params[:, 4].astype(int)) & (r["mirror"] == params[:, 3].astype(int)) & (r["angle_bin"] == jt)
    da = np.abs(((r["alpha"][ok] - params[ok, 0]) + 180.0) % 360.0 - 180.0)
    rec = {"workload": name + (" / quadri" if interp else "") + ("" if normalize is None else " / Normalize_ring %s" % ("on" if normalize else "off")), "geometry": {"nx": nx, "ou": ou, "xr": xr, "ts": ts, "nref": refs_n.shape[0]}, "particles": n, "sigma": sigma,
           "search_path": bench.SEARCH_PATHS[path] + (" (tiled)" if tiled else ""), "tie_flips": flips, "flip_classes": kinds,
           "max_rel_peak": float(rel.max()), "alpha_outliers_gt_2e-3_deg": int((da > 2e-3).sum()),
           "max_alpha_diff_deg": float(da.max()) if da.size else 0.0, "refined_by_exact_kernel": refined,
           "oracle_seconds": round(t_or, 1), "oracle_threads": threads}
    print(json.dumps(rec), flush=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="*", default=["mref", "reffree_iter0", "reffree_aligned", "mref50"])
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_audit_large.json"))
    ap.add_argument("--interp", type=int, default=0, help="1: Util::quadri in alrl_ms (engine option RA_INTERP_QUADRI, oracle ORC_INTERP_QUADRI)")
    ap.add_argument("--normalize-ring", type=int, default=-1, help="0 / 1: Normalize_ring off / on whatever the mode (engine option; oracle flag)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    threads = bench.host_cores()
    recs = []
    for w in a.workloads:
        n = min(a.n, CAPS.get(w, a.n))
        recs.append(run_case(w, n, a.sigma, dev, threads, a.interp, None if a.normalize_ring < 0 else bool(a.normalize_ring)))
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump({"_what": "engine against the oracle on large sigma = %g samples: disagreements of the integer assignment by class "
                                "(scripts/parity_audit_large.py)" % a.sigma, "cases": recs}, f, indent=1)


if __name__ == "__main__":
    main()
