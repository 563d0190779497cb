#!/usr/bin/env python3
"""bench.py -- particles/s of the multi-reference 2-D alignment hot path on MI355X.

One "step" = one alignment iteration over the resident particle shard: reference preparation,
search over (reference, shift, mirror, angle), rot_shift2D + per-class even/odd accumulation,
the class-sum all-reduce and the reference update (test_mref_gpu_align.py:408-575).
Workload at N=1: BASELINE.json configs[1] -- 50 000 synthetic 90x90 particles, nref=10,
xr=yr=3, ts=1, ou=36.  Particles are already resident in HBM when the timed region starts.
For N>1 every rank holds its own 50 000-particle shard (weak scaling) and the only collective
is the RCCL all-reduce of the class sums.

    python bench.py --gpus 1 --steps 6 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cryo_ralib_amd import api, dist as rdist, geometry, synth  # noqa: E402
from cryo_ralib_amd.mref import MrefAligner  # noqa: E402

PEAK_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: f32 MFMA == f32 vector peak


def algorithmic_flops(nx, ou, xr, yr, ts, nref):
    """SURVEY.md §8(d) work formula, split into the polar/FFT stage and the CCF stage."""
    numr = geometry.numrinit(1, ou, 1)
    lens = numr[2::3]
    L = sum(lens)
    M = numr[-1]
    S = (2 * int(xr / ts) + 1) * (2 * int(yr / ts) + 1)
    polar = S * (8 * L + 5 * L + sum(2.5 * n * math.log2(n) for n in lens))
    ccf = S * nref * (12 * (L / 2) + 2 * 2.5 * M * math.log2(M))
    return polar, ccf, S, L, M


def generate_shard(engine_dev, refs_np, n, xr, yr, sigma, shard, nx, ou):
    """synthetic particles built ON the GPU with the product's own rot_shift2D kernel
    (rotate -> shift -> mirror of a random class reference) + Gaussian noise; BASELINE.md §3."""
    nref = refs_np.shape[0]
    truth = synth.plant_truth(nref, n, xr, yr, 2000 + shard)
    gen = api.Engine(nx, ou, xr, yr, 1.0, nref, api.RA_MODE_MREF, device=engine_dev.index)
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
    step = 8192
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


def cpu_baseline(refs_np, nx, ou, xr, yr, nref, target_seconds=15.0):
    """the CPU restatement of the EMAN2 path (oracle/, kind "port") timed on this host's cores
    on a bounded sample of the same workload."""
    from oracle import oracle as orc
    threads = host_cores()
    rg = orc.rings(1, ou, 1)
    mask = orc.model_circle(ou, nx, nx)
    _, cref = orc.prepare_refs(refs_np, mask, rg)
    n0 = 4 * threads
    parts, _ = synth.make_particles(refs_np, n0, xr, yr, 1.0, shard=99, ou=ou)
    d = np.zeros((n0, 2), np.float32)
    t = time.perf_counter()
    orc.mref_iteration(parts, cref, rg, xr, yr, 1.0, d, nthreads=threads)
    rate = n0 / (time.perf_counter() - t)
    n1 = int(max(n0, min(20000, rate * target_seconds)))
    reps = (n1 + n0 - 1) // n0
    big = np.concatenate([parts] * reps)[:n1]
    d = np.zeros((n1, 2), np.float32)
    t = time.perf_counter()
    orc.mref_iteration(big, cref, rg, xr, yr, 1.0, d, nthreads=threads)
    dt = time.perf_counter() - t
    return {"value": n1 / dt, "unit": "particles/s", "cores": threads, "kind": "port",
            "sample": "%d particles x 1 iteration of the oracle's mref_ali2d loop (search + rot_shift2D + class sums), "
                      "%d OpenMP threads, %.1f s" % (n1, threads, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--particles", type=int, default=50000, help="particles per GPU")
    ap.add_argument("--nref", type=int, default=10)
    ap.add_argument("--nx", type=int, default=90)
    ap.add_argument("--ou", type=int, default=36)
    ap.add_argument("--xr", type=float, default=3.0)
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--function", default="ref_ali2d", help="reference preparation per iteration: ref_ali2d (the "
                    "reference's default --function: FSC-fitted tangent filter + centring, on the device) | none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank, local, world = rdist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the alignment engine has no CPU path")
    local = local % max(torch.cuda.device_count(), 1)      # rehearsal: more ranks than GPUs share devices
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    nx, ou, xr, nref, n = args.nx, args.ou, args.xr, args.nref, args.particles

    refs_np = synth.make_references(nref, nx, ou)
    particles, _ = generate_shard(dev, refs_np, n, xr, xr, args.sigma, rank, nx, ou)
    # the run starts from the generating references; every step re-estimates them from the data
    al = MrefAligner(particles, refs_np, ou, xr, xr, 1.0, device=local, index0=rank * n, total_nima=n * world,
                     preprocess=True, chunk=args.chunk)
    user_func = None if args.function in ("none", "None", "") else args.function
    for _ in range(args.warmup):
        al.iterate(user_func, 1)
    al.engine.kernel_time(True)          # arm HIP-event timing of the hot kernels on the engine's stream
    rdist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        al.iterate(user_func, 1)
    rdist.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = rdist.max_over_ranks(dt, dev)
    ms_ccf, n_ccf, ms_polar, n_polar = al.engine.kernel_time(False)

    if rank == 0:
        # HBM bytes of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 on
        # gfx950 + WRITE_SIZE, per particle), scaled to this run's particles per launch
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_v8_pmc_summary.json")) as f:
                pm = json.load(f)
            if (nx, ou, nref) == (90, 36, 10):
                key = [k for k in pm["kernels"] if "ccf_kernel<256>" in k][0]
                per_particle = pm["kernels"][key]["hbm_bytes_per_dispatch_corrected"] / pm["particles_per_dispatch"]
                traffic = per_particle * (n * args.steps / max(n_ccf, 1))
        except (OSError, KeyError, ValueError, IndexError):
            pass
        total = n * world * args.steps
        polar_f, ccf_f, S, L, M = algorithmic_flops(nx, ou, xr, xr, 1.0, nref)
        # average launch: total particles through the kernel / launches
        part_per_launch = n * args.steps / max(n_ccf, 1)
        avg_ms = ms_ccf / max(n_ccf, 1)
        achieved = ccf_f * part_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        line = {
            "metric": "particles/sec aligned, %dx%d nref=%d xr=yr=%g ou=%d" % (nx, nx, nref, xr, ou),
            "value": total / dt, "unit": "particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("%s: %d synthetic %dx%d particles per GPU, nref=%d, xr=yr=%g, ts=1, "
                                    "ou=%d; step = one mref_ali2d iteration (search + rot_shift2D + class sums + "
                                    "all-reduce + reference update with --function=" + str(args.function) + ")") % (
                                       "BASELINE configs[1]" if (nx, ou, nref, xr) == (90, 36, 10, 3.0) else "custom", n, nx, nx, nref, xr, ou),
                       "particles_per_gpu": n, "nref": nref, "shifts": S, "parallelism": "dp%d" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_TFLOPS, "traffic": traffic,
                         "kernel": "ccf_kernel<%d> (Crosrng_ms contraction + IFFT + argmax)" % M,
                         "flops_per_particle": ccf_f, "particles_per_launch": part_per_launch,
                         "avg_launch_ms": avg_ms, "launches": n_ccf,
                         "polar_fft_kernel": {"avg_launch_ms": ms_polar / max(n_polar, 1), "flops_per_particle": polar_f}},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(refs_np, nx, ou, xr, xr, nref)
        print(json.dumps(line))
    al.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
