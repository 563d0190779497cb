#!/usr/bin/env python3
"""bench.py -- particles/s of the 2-D alignment hot path on MI355X.

One "step" = one alignment iteration over the resident particle shard: reference preparation, search over
(reference, shift, mirror, angle), rot_shift2D + per-class even/odd accumulation, the class-sum all-reduce and the
reference update (test_mref_gpu_align.py:408-575; test_reffree_gpu_align.py:361-540 for --workload reffree).
Particles are already resident in HBM when the timed region starts.  For N>1 every rank holds its own shard (weak
scaling) and the only collective is the RCCL all-reduce of the class sums.

    python bench.py --gpus 1 --steps 6 --warmup 1                       # BASELINE configs[1] (the metric's config); at one GPU
                                                                        # the line also carries short runs of the three workloads
                                                                        # below under "other_workloads" (--no-others: headline only)
    python bench.py --workload reffree                                  # BASELINE configs[2]
    python bench.py --workload largebox                                 # BASELINE configs[4] geometry, one GPU's share
    python bench.py --workload mref50                                   # BASELINE configs[3], one GPU's share (125k particles, nref=50)
    python bench.py --workload nb00 | box128 | box100                   # boxes beyond the 90 x 90 kernels: the reference notebook's 130 x 130 / ou 52 /
                                                                        # nref 50, 128 x 128 / ou 60, 100 x 100 / ou 40
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` without a torch.distributed launcher re-launches itself under one (N child processes) before any GPU call;
a launcher whose world size differs from N is an error.  It never silently runs a single rank.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: f32 MFMA == f32 vector peak
PEAK_HBM_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E ~8 TB/s
N_SIMD = 256 * 4            # MI355X: 256 CUs x 4 SIMDs (matrix-pipe utilisation from SQ_VALU_MFMA_BUSY_CYCLES)

SEARCH_PATHS = {0: "kernel pair", 1: "fused", 2: "generic", 3: "solo / duo / pair (particle-resident, one or two ring buffers next to the image)"}

WORKLOADS = {
    # name: (BASELINE.json config, nx, ou, xr, nref, particles per GPU, default steps, warmup)
    "mref": ("configs[1]", 90, 36, 3.0, 10, 50000, 6, 1),
    "reffree": ("configs[2]", 90, 36, 3.0, 1, 50000, 10, 1),
    "largebox": ("configs[4] geometry, one GPU's share", 256, 120, 5.0, 100, 8192, 3, 1),
    "mref50": ("configs[3], one GPU's share", 90, 36, 3.0, 50, 125000, 4, 1),
    # rings of 512 samples (search_solo_kernel): the reference's own documented run (notebook/00_Multireference_Alignment.ipynb
    # cell 3: 5000 x 130 x 130, nref = 50, ou = 52) and a 128 x 128 box at the largest radius
    # (two warm-up steps: in a process that ran another workload before, one of the first two iterations of these short runs takes
    # 50 - 90 ms longer -- the driver unmapping the gigabytes the previous workload freed --, scripts/dev/two_workloads2.py)
    "nb00": ("reference notebook/00 cell 3 geometry", 130, 52, 3.0, 50, 5000, 6, 2),
    "box128": ("128 x 128 box, ou = 60 (maxrin 512)", 128, 60, 3.0, 10, 16384, 6, 2),
    # rings of 256 samples in a box too large for four LDS ring buffers (search_pair_kernel)
    "box100": ("100 x 100 box, ou = 40 (maxrin 256, two ring buffers)", 100, 40, 3.0, 10, 32768, 6, 2),
    # a box far larger than the rings: the LDS image is a crop around the particle's centre (search_pair_kernel), rot_shift2D +
    # class sums by output tiles (transform_sum_tile_kernel)
    "box256": ("256 x 256 box, ou = 36 (cropped LDS image)", 256, 36, 3.0, 10, 8192, 6, 2),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="mref")
    ap.add_argument("--particles", type=int, default=None, help="particles per GPU")
    ap.add_argument("--nref", type=int, default=None)
    ap.add_argument("--nx", type=int, default=None)
    ap.add_argument("--ou", type=int, default=None)
    ap.add_argument("--xr", type=float, default=None)
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--function", default="ref_ali2d", help="reference preparation per iteration: ref_ali2d (the "
                    "reference's default --function: FSC-fitted tangent filter + centring, on the device) | none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-pcie", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="headline workload only (the default run appends short runs of "
                    "reffree, mref50 and largebox under \"other_workloads\")")
    args = ap.parse_args(argv)
    cfg, nx, ou, xr, nref, n, steps, warm = WORKLOADS[args.workload]
    args.config_name = cfg
    args.custom_geometry = any(v is not None for v in (args.particles, args.nref, args.nx, args.ou, args.xr)) or args.chunk != 0
    args.nx = args.nx or nx; args.ou = args.ou or ou; args.xr = xr if args.xr is None else args.xr
    args.nref = args.nref or nref; args.particles = args.particles or n
    args.steps = steps if args.steps is None else args.steps
    args.warmup = warm if args.warmup is None else args.warmup
    return args


def relaunch_under_launcher(args):
    """--gpus N given to a plain `python bench.py`: start N ranks as child processes (before this process touches
    the GPU) and hand back their exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def algorithmic_flops(nx, ou, xr, yr, ts, nref):
    """SURVEY.md section 8(d) work formula, split into the polar/FFT stage and the CCF stage."""
    from cryo_ralib_amd import geometry
    numr = geometry.numrinit(1, ou, 1)
    lens = numr[2::3]
    L = sum(lens)
    M = numr[-1]
    S = (2 * int(xr / ts) + 1) * (2 * int(yr / ts) + 1)
    polar = S * (8 * L + 5 * L + sum(2.5 * n * math.log2(n) for n in lens))
    ccf = S * nref * (12 * (L / 2) + 2 * 2.5 * M * math.log2(M))
    return polar, ccf, S, L, M


def live_offsets(state, nx, ou, xr, yr, ts, reffree):
    """search offsets inside every particle's window (sp_alignment.search_range after the reset / clamp rule,
    test_mref_gpu_align.py:1030-1038): the solo kernel skips the others, so its algorithmic work is counted per live offset.
    state: [n][2] device tensor; returns a 0-d device tensor (mean live offsets per particle)."""
    import torch
    cn = nx // 2 + 1
    mashi = float(cn - ou - 2)
    if reffree:
        d = state.clamp(-mashi, mashi)
    else:
        d = torch.where((state.abs() > mashi).any(1, keepdim=True), torch.zeros_like(state), state)
    cnt = []
    for ax, rng in ((0, xr), (1, yr)):
        ql = (cn + d[:, ax] - ou - 2).clamp(min=0.0).clamp(max=rng)
        qe = (nx - cn - d[:, ax] - ou).clamp(min=0.0).clamp(max=rng)
        cnt.append(torch.floor(ql / ts) + torch.floor(qe / ts) + 1.0)
    return (cnt[0] * cnt[1]).mean()


def generate_shard(engine_dev, refs_np, n, xr, yr, sigma, shard, nx, ou):
    """synthetic particles built ON the GPU with the product's own rot_shift2D kernel
    (rotate -> shift -> mirror of a random class reference) + Gaussian noise; BASELINE.md section 3."""
    import numpy as np
    import torch
    from cryo_ralib_amd import api, synth
    nref = refs_np.shape[0]
    truth = synth.plant_truth(nref, n, xr, yr, 2000 + shard)
    gen = api.Engine(nx, ou, xr, yr, 1.0, min(nref, 8), api.RA_MODE_MREF, device=engine_dev.index)
    gen.use_current_stream()
    refs = torch.from_numpy(refs_np).to(engine_dev)
    out = torch.empty((n, nx, nx), dtype=torch.float32, device=engine_dev)
    g = torch.Generator(device=engine_dev)
    g.manual_seed(int(truth["noise_seed"]))
    rec = np.zeros(n, api.RESULT_DTYPE)
    rec["alpha"] = truth["ang"]; rec["sx"] = truth["sx"]; rec["sy"] = truth["sy"]
    rec["mirror"] = truth["mir"]; rec["ref_id"] = truth["cls"]
    res = torch.from_numpy(rec.view(np.int32).reshape(n, 8)).to(engine_dev)
    cls = torch.from_numpy(truth["cls"].astype(np.int64)).to(engine_dev)
    step = 8192 if nx <= 128 else 512
    for s in range(0, n, step):
        e = min(n, s + step)
        src = refs[cls[s:e]].contiguous()
        gen.transform_accumulate(src, res[s:e].contiguous(), 0, out[s:e], None, None)
        out[s:e] += torch.randn((e - s, nx, nx), generator=g, device=engine_dev) * sigma
    gen.sync()
    gen.close()
    return out, truth


def host_cores():
    """CPU share of this process: cgroup quota if one is set, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(refs_np, nx, ou, xr, yr, nref, reffree, target_seconds=10.0):
    """the CPU restatement of the EMAN2 path (oracle/, kind "port") timed on this host's cores on a bounded sample
    of the same workload: all cores (OpenMP over particles), and one thread (what one EMAN2 MPI rank does)."""
    import numpy as np
    from cryo_ralib_amd import synth
    from oracle import oracle as orc
    threads = host_cores()
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    _, cref = orc.prepare_refs(refs_np, mask if not reffree else None, rg)
    n0 = 4 * threads if nx <= 128 else threads
    parts, _ = synth.make_particles(refs_np, n0, xr, yr, 1.0, shard=99, ou=ou)

    def run(p, nthreads):
        d = np.zeros((p.shape[0], 2), np.float32)
        t = time.perf_counter()
        if reffree:
            orc.reffree_iteration(p, cref[0], rg, xr, yr, 1.0, (0, 0), d, np.zeros((p.shape[0], 6), np.float32), nthreads=nthreads)
        else:
            orc.mref_iteration(p, cref, rg, xr, yr, 1.0, d, nthreads=nthreads)
        return time.perf_counter() - t

    rate = n0 / run(parts, threads)
    n1 = int(max(n0, min(40000, rate * target_seconds)))
    big = np.concatenate([parts] * ((n1 + n0 - 1) // n0))[:n1]
    runs = sorted(run(big, threads) for _ in range(1 if nx > 128 else 3))
    dt = runs[len(runs) // 2]
    n2 = int(max(4, min(n1, rate / threads * 6.0)))
    dt1 = run(big[:n2], 1)
    what = "ali2d_single_iter" if reffree else "mref_ali2d"
    return {"value": n1 / dt, "unit": "particles/s", "cores": threads, "kind": "port",
            "sample": "%d particles x 1 iteration of the oracle's %s loop (search + rot_shift2D + class sums), "
                      "%d OpenMP threads, %.1f s (median of %d)" % (n1, what, threads, dt, len(runs)),
            "single_thread": {"value": n2 / dt1, "unit": "particles/s", "cores": 1,
                              "sample": "%d particles x 1 iteration, 1 thread, %.1f s" % (n2, dt1)}}


def parity_block(refs_np, nx, ou, xr, nref, reffree, dev, n=4096):
    """SURVEY.md section 8(d) "parity checks reported with every perf number": the engine against the oracle on a
    sigma = 0.25 and a sigma = 1.0 subsample of the same synthetic workload."""
    import numpy as np
    import torch
    from cryo_ralib_amd import api
    from oracle import oracle as orc
    if nx > 130:
        # large boxes: 32 per set where the rings reach 1024 samples (configs[4]: the 16-thread oracle takes ~10 s for them), 256
        # where they stay short (256 x 256 / ou = 36)
        n = 32 if ou > 64 else 256
    elif nx > 90:
        n = 512           # maxrin 512: ~1 k particles/s on 16 threads at nref = 10
    elif nref > 16:
        n = 1024          # the oracle's share of the run grows with the reference count
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    out = {"particles_per_set": n, "peak_tolerance": 1e-4}
    for sigma in (0.25, 1.0):
        parts_t, _ = generate_shard(dev, refs_np, n, xr, xr, sigma, 7, nx, ou)
        parts = parts_t.cpu().numpy()
        parts = np.stack([orc.normalize_mask(p, mask, 0) for p in parts])
        d = np.zeros((n, 2), np.float32)
        if reffree:
            tavg = parts.mean(0)[None].astype(np.float32)
            refs_n, cref = orc.prepare_refs(tavg, None, rg)
            params, infos, _, _ = orc.reffree_iteration(parts, cref[0], rg, xr, xr, 1.0, (0, 0), d, np.zeros((n, 6), np.float32),
                                                        nthreads=host_cores())
            mode = api.RA_MODE_REFFREE
        else:
            refs_n, cref = orc.prepare_refs(refs_np, mask, rg)
            params, infos, _, _ = orc.mref_iteration(parts, cref, rg, xr, xr, 1.0, d, nthreads=host_cores())
            mode = api.RA_MODE_MREF
        eng = api.Engine(nx, ou, xr, xr, 1.0, refs_n.shape[0], mode, device=dev.index)
        eng.set_references(torch.from_numpy(np.ascontiguousarray(refs_n)).to(dev))
        st, res = eng.new_state(n), eng.new_result(n)
        eng.align(torch.from_numpy(parts).to(dev), st, res)
        eng.sync()
        r = eng.result_to_numpy(res)
        path = eng.search_path
        eng.close()
        jt = np.array([infos[i].jtot for i in range(n)])
        same = (r["ref_id"] == params[:, 4].astype(int)) & (r["mirror"] == params[:, 3].astype(int)) & (r["angle_bin"] == jt) & \
               (np.abs(st.cpu().numpy() - d).max(1) < 1e-6)
        rel = np.abs(r["peak"] - params[:, 5]) / np.abs(params[:, 5])
        # sub-bin angle (prb1d on the 7-point neighbourhood of the peak) of the particles whose integer assignment agrees
        da = np.abs(((r["alpha"][same] - params[same, 0]) + 180.0) % 360.0 - 180.0)
        key = "sigma_%g" % sigma
        out[key] = {"max_rel_peak": float(rel.max()), "exact_match_rate": float(same.mean()),
                    "tie_flips": int((~same).sum()),
                    "max_rel_peak_gap_of_flips": float(rel[~same].max()) if (~same).any() else 0.0,
                    "alpha_outliers_gt_2e-3_deg": int((da > 2e-3).sum()), "max_alpha_diff_deg": float(da.max()) if da.size else 0.0}
        out["search_path"] = SEARCH_PATHS[path]
    return out


def committed_traffic(kernel_substr, workload, geometry=None):
    """HBM bytes per particle of the dominant kernel from the committed rocprofv3 --pmc passes of this workload
    (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md HBM section; scripts/profile.sh); PMC counters cannot be
    read from inside this process."""
    import glob
    import re
    if geometry is not None and tuple(geometry) != tuple(WORKLOADS[workload][1:5]):
        return None, None, None    # the summaries were profiled at the workload's default (nx, ou, xr, nref) only
    natural = lambda q: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(q))]      # r02_v10 after r02_v9
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), key=natural, reverse=True):
        try:
            with open(path) as f:
                pm = json.load(f)
            wl = "mref"
            for w in WORKLOADS:
                if "--workload " + w in pm.get("bench_args", ""):
                    wl = w
            if wl != workload or not pm.get("particles_per_dispatch"):
                continue
            # all dispatches of the kernel(s) over the 14 000 particles of a PMC pass (scripts/profile.sh); the large-box contraction
            # is two kernels per slice since round 4 (contraction + inverse transforms), timed together by the engine
            names = (kernel_substr, "gccf_ifft_kernel") if kernel_substr == "ccf_generic_kernel" else (kernel_substr,)
            tot = sum(v["hbm_bytes_per_dispatch_corrected"] * v.get("dispatches", 1) for k, v in pm["kernels"].items()
                      if any(nm in k for nm in names) and "pack" not in k and "hbm_bytes_per_dispatch_corrected" in v)
            if tot > 0:
                # matrix-pipe utilisation of the dominant kernel from the same passes: SQ_VALU_MFMA_BUSY_CYCLES (summed over the
                # SIMDs) / (kernel cycles x SIMDs); GRBM_GUI_ACTIVE counts the kernel's cycles once per XCD (8)
                busy = None
                ks = [v for k, v in pm["kernels"].items() if kernel_substr in k and "pack" not in k and v.get("GRBM_GUI_ACTIVE")]
                if ks:
                    cyc = sum(v["GRBM_GUI_ACTIVE"] * v.get("dispatches", 1) for v in ks) / 8.0
                    busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * v.get("dispatches", 1) for v in ks) / (cyc * N_SIMD)
                return tot / 14000.0, os.path.relpath(path, ROOT), busy
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            continue
    return None, None, None


def run_workload(args, rank, local, world, dev):
    """one workload: generate the shard, warm up, time `args.steps` steps; returns the JSON line (rank 0) or None"""
    import numpy as np
    import torch
    from cryo_ralib_amd import api, dist as rdist, synth
    from cryo_ralib_amd.mref import MrefAligner, RefFreeAligner

    nx, ou, xr, nref, n = args.nx, args.ou, args.xr, args.nref, args.particles
    reffree = args.workload == "reffree"
    user_func = None if args.function in ("none", "None", "") else args.function

    refs_np = synth.make_references(max(nref, 1), nx, ou)
    particles, _ = generate_shard(dev, refs_np, n, xr, xr, args.sigma, rank, nx, ou)
    pcie = None
    if rank == 0 and not args.no_pcie:
        # the boundary hands over host buffers (pre_align_fetch): one pinned H2D copy of the shard, measured once
        host = torch.empty(particles.shape, dtype=torch.float32, pin_memory=True)
        host.copy_(particles)
        torch.cuda.synchronize()
        t = time.perf_counter()
        particles.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        pcie = time.perf_counter() - t
        del host
    # the run starts from the generating references; every step re-estimates them from the data
    if reffree:
        al = RefFreeAligner(particles, ou, xr, xr, 1.0, device=local, index0=rank * n, total_nima=n * world, preprocess=True,
                            chunk=args.chunk)
        step = lambda: al.iterate(-1, user_func)
    else:
        al = MrefAligner(particles, refs_np, ou, xr, xr, 1.0, device=local, index0=rank * n, total_nima=n * world,
                         preprocess=True, chunk=args.chunk)
        step = lambda: al.iterate(user_func, 1)
    for _ in range(args.warmup):
        step()
    al.engine.kernel_time(True)          # arm HIP-event timing of the hot kernels on the engine's stream
    rdist.barrier(); torch.cuda.synchronize()
    solo = al.engine.search_skips_offsets          # the kernels evaluate in-window offsets only: count the work that is asked for AND done
    live = torch.zeros((), device=dev)
    if solo:
        live_offsets(al.state, nx, ou, xr, xr, 1.0, reffree)      # first use of these torch operators (their one-off set-up) stays outside the timed region
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if solo:          # a few small device-side operations per step, no synchronisation: the windows of the coming search
            live += live_offsets(al.state, nx, ou, xr, xr, 1.0, reffree)
        step()
    rdist.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = rdist.max_over_ranks(dt, dev)
    ms_a, n_a, ms_b, n_b = al.engine.kernel_time(False)
    path = al.engine.search_path
    tiled = al.engine.search_tiled
    opp = al.engine.search_offsets_per_pass

    if rank == 0:
        total = n * world * args.steps
        polar_f, ccf_f, S, L, M = algorithmic_flops(nx, ou, xr, xr, 1.0, nref)
        live_frac = 1.0
        if solo:
            # the solo / duo / pair kernels and the size-generic kernels (live-offset lists) compute the in-window offsets only (a box
            # that is tight around the rings -- ou = 60 in 128 pixels, ou = 120 in 256 -- leaves shifted particles a fraction of their
            # offsets): count the work that was asked for AND done
            live_frac = float(live.item()) / args.steps / S
            polar_f *= live_frac; ccf_f *= live_frac
        per_launch = n * args.steps / max(n_a, 1)
        kernels = {}
        if path == 3:
            kernels[("search_pair_kernel<%d>" if M == 256 else "search_duo_kernel<%d>" if opp == 2 else "search_solo_kernel<%d>") % M] = {
                "what": "polar resampling + Normalize_ring + ring FFT + Crosrng_ms contraction (4x4x1 MFMA) + %d-point inverse FFT + argmax, "
                        "particle-resident, %s, reference tiles with the A operand in registers" %
                        (M, "two search offsets per pass in two LDS ring buffers" if M == 256 else
                         "two search offsets per pass through one LDS ring buffer" if opp == 2 else "one search offset per pass"),
                "avg_launch_ms": ms_a / max(n_a, 1), "launches": n_a, "flops_per_particle": polar_f + ccf_f}
        elif path == 1:
            kernels[("search_tiled_kernel<%d>" if tiled else "search_fused_kernel<%d>") % M] = {
                "what": "polar resampling + Normalize_ring + ring FFT + Crosrng_ms contraction (4x4x1 MFMA) + inverse FFT + argmax, "
                        "particle-resident" + (", reference tiles of <= 10 with the A operand in registers" if tiled else ""), "avg_launch_ms": ms_a / max(n_a, 1), "launches": n_a, "flops_per_particle": polar_f + ccf_f}
        else:
            nm = ("ccf_generic_kernel", "polar_generic_kernel") if path == 2 else ("ccf_kernel<%d>" % M, "polar_fft_kernel")
            kernels[nm[0]] = {"what": "Crosrng_ms contraction (16x16x4 MFMA) + inverse FFT + argmax" +
                                      (" (maxrin 1024: ccf_generic_kernel<SPLIT> + gccf_ifft_kernel per slice of 512 blocks, timed together per chunk)" if path == 2 and M == 1024 else ""),
                              "avg_launch_ms": ms_a / max(n_a, 1),
                              "launches": n_a, "flops_per_particle": ccf_f}
            kernels[nm[1]] = {"what": "Polar2Dm + Normalize_ring + Frngs", "avg_launch_ms": ms_b / max(n_b, 1), "launches": n_b,
                              "flops_per_particle": polar_f}
        for k in kernels.values():
            k["achieved_tflops"] = k["flops_per_particle"] * per_launch / (k["avg_launch_ms"] * 1e-3) / 1e12 if k["avg_launch_ms"] > 0 else 0.0
            k["frac"] = k["achieved_tflops"] / PEAK_F32_TFLOPS
        dom = max(kernels, key=lambda k: kernels[k]["avg_launch_ms"] * kernels[k]["launches"])
        kern_ms = sum(k["avg_launch_ms"] * k["launches"] for k in kernels.values())
        traffic_pp, traffic_src, mfma_busy = committed_traffic(dom.split("<")[0], args.workload, (nx, ou, xr, nref))
        whole = (polar_f + ccf_f) * total / world / dt / 1e12
        # Roofline of the dominant kernel, both roofs on ALGORITHMIC work: flops per particle against the f32
        # matrix peak, bytes per particle (image read for the search + read for rot_shift2D + 24 B of parameters) against HBM; the
        # larger fraction names the bound.  What the kernel really moves through the L2 <-> fabric boundary (committed PMC bytes,
        # Infinity Cache hits included) is reported beside it as `traffic` and `hbm_bus_frac` -- a bus utilisation, not a fraction
        # of useful work: the large-box kernels move ~690 x their algorithmic bytes.
        alg_bytes = 2 * nx * nx * 4 + 24
        launch_s = kernels[dom]["avg_launch_ms"] * 1e-3
        hbm_alg_gbps = alg_bytes * per_launch / launch_s / 1e9 if launch_s > 0 else 0.0
        hbm_alg_frac = hbm_alg_gbps / PEAK_HBM_GBPS
        # (the PMC passes run a first iteration -- every offset inside its window; kernels that skip out-of-window offsets move
        # bytes in proportion to the live ones, so their committed figure is scaled by the live fraction of the timed steps)
        traffic_full = traffic_pp
        if traffic_pp is not None and solo:
            traffic_pp = traffic_pp * live_frac
        hbm_gbps = traffic_pp * per_launch / launch_s / 1e9 if traffic_pp is not None and launch_s > 0 else None
        hbm_frac = hbm_gbps / PEAK_HBM_GBPS if hbm_gbps is not None else None
        hbm_bound = hbm_alg_frac > kernels[dom]["frac"]
        what = ("%s: %d synthetic %dx%d particles per GPU, %s, xr=yr=%g, ts=1, ou=%d; step = one %s iteration (search + rot_shift2D + "
                "class sums + all-reduce + reference update with --function=%s)") % (
                   "BASELINE " + args.config_name if (args.nx, args.ou, args.xr, args.nref) == WORKLOADS[args.workload][1:5] else "custom",
                   n, nx, nx, "reference-free (running average)" if reffree else "nref=%d" % nref, xr, ou,
                   "ali2d" if reffree else "mref_ali2d", args.function)
        line = {
            "metric": "particles/sec aligned, %dx%d %s xr=yr=%g ou=%d" % (nx, nx, "reference-free ali2d" if reffree else "nref=%d" % nref, xr, ou),
            "value": total / dt, "unit": "particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": what, "particles_per_gpu": n, "nref": nref, "shifts": S, "live_shift_fraction": live_frac, "parallelism": "dp%d" % world,
                       "search_path": SEARCH_PATHS[path]},
            "roofline": {"bound": "hbm" if hbm_bound else "mfma",
                         "achieved": hbm_alg_gbps if hbm_bound else kernels[dom]["achieved_tflops"],
                         "peak": PEAK_HBM_GBPS if hbm_bound else PEAK_F32_TFLOPS,
                         "unit": "GB/s" if hbm_bound else "TFLOP/s",
                         "frac": hbm_alg_frac if hbm_bound else kernels[dom]["frac"],
                         "mfma_frac": kernels[dom]["frac"], "hbm_algorithmic_gbps": hbm_alg_gbps, "hbm_algorithmic_frac": hbm_alg_frac,
                         "mfma_busy_frac": mfma_busy,
                         "mfma_busy_what": "matrix-pipe utilisation of this kernel in the committed PMC passes: SQ_VALU_MFMA_BUSY_CYCLES / "
                                           "(kernel cycles x 1024 SIMDs)",
                         "hbm_bus_gbps": hbm_gbps, "hbm_bus_frac": hbm_frac,
                         "hbm_bus_what": "measured bytes through the L2 <-> fabric boundary (traffic) / live launch time / 8 TB/s: bus "
                                         "utilisation, not a roofline fraction of useful work",
                         "traffic": traffic_pp * per_launch if traffic_pp is not None else None,
                         "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x 2 + WRITE_SIZE per particle x particles per launch; x live_shift_fraction "
                                         "for kernels that evaluate in-window offsets only: the PMC passes run full windows)",
                         "traffic_per_particle": traffic_pp, "traffic_per_particle_full_windows": traffic_full, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_particle": alg_bytes,
                         "kernel": dom, "flops_per_particle": kernels[dom]["flops_per_particle"],
                         "particles_per_launch": per_launch, "avg_launch_ms": kernels[dom]["avg_launch_ms"],
                         "launches": kernels[dom]["launches"], "kernels": kernels,
                         "hot_kernels_share_of_step": kern_ms / (dt * 1e3),
                         "whole_path": {"achieved": whole, "frac": whole / PEAK_F32_TFLOPS,
                                        "flops_per_particle": polar_f + ccf_f}},
        }
        if pcie is not None:
            line["pcie_inclusive"] = {"value": total / world / (pcie + dt), "unit": "particles/s per GPU",
                                      "h2d_seconds": pcie, "h2d_gb_per_s": particles.numel() * 4 / pcie / 1e9,
                                      "note": "one pinned H2D copy of the shard + the %d timed steps" % args.steps}
        if world == 1 and not args.no_parity:
            line["parity"] = parity_block(refs_np, nx, ou, xr, nref, reffree, dev)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(refs_np, nx, ou, xr, xr, nref, reffree)
    al.close()
    del al, particles
    torch.cuda.empty_cache()
    return line if rank == 0 else None


# what the default run adds to the headline line: the other BASELINE configs at one GPU's share, a few steps each
OTHER_WORKLOADS = ("reffree", "mref50", "largebox", "nb00", "box128", "box100", "box256")


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(relaunch_under_launcher(args))
    if world_env is not None and int(world_env) != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %s ranks" % (args.gpus, world_env))

    import torch
    from cryo_ralib_amd import dist as rdist

    rank, local, world = rdist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the alignment engine has no CPU path")
    local = local % max(torch.cuda.device_count(), 1)      # rehearsal: more ranks than GPUs share devices
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        # one line per rank on stderr: which device this rank bound and over what backend, so that the first run on a whole node
        # diagnoses itself (two ranks on one ordinal, a rank without a GPU, gloo instead of RCCL)
        import torch.distributed as tdist
        props = torch.cuda.get_device_properties(local)
        print("bench.py rank %d/%d: LOCAL_RANK %s -> cuda:%d %s (%d CUs, %.0f GiB), %d device(s) visible, backend %s" % (
            rank, world, os.environ.get("LOCAL_RANK", "?"), local, torch.cuda.get_device_name(local), props.multi_processor_count,
            props.total_memory / 2.0 ** 30, torch.cuda.device_count(), tdist.get_backend() if tdist.is_initialized() else "none"),
            file=sys.stderr, flush=True)
    line = run_workload(args, rank, local, world, dev)
    if world == 1 and args.workload == "mref" and not args.no_others and not args.custom_geometry:
        # the other workloads, measured in the same process right after the headline (short runs, their own parity block, no
        # CPU baseline); the headline keys above are untouched
        others = {}
        for w in OTHER_WORKLOADS:
            sub = parse_args(["--workload", w, "--no-cpu-baseline", "--no-pcie", "--sigma", str(args.sigma)] +
                             (["--no-parity"] if args.no_parity else []))
            t0 = time.perf_counter()
            o = run_workload(sub, rank, local, world, dev)
            o = {k: o[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "parity") if k in o}
            o["roofline"] = {k: v for k, v in o["roofline"].items() if k != "kernels"}
            o["wall_s"] = time.perf_counter() - t0
            others[w] = o
        line["other_workloads"] = others
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
