"""Command-line entry points with the reference's arguments.

  test_mref_gpu_align.py    data_stack reference_stack outdir <maskfile> --ir --ou --rs --xr --yr --ts
                            --center --maxit --CTF --snr --function --rand_seed --gpu_devices --gpu_info --MPI --EQ
                            (test_mref_gpu_align.py:1136-1160)
  test_reffree_gpu_align.py data_stack outdir <maskfile> --ir --ou --rs --xr --yr --ts --maxit --center ...
                            (test_reffree_gpu_align.py:911-985)

One process per GPU: run under `python -m torch.distributed.run --nproc-per-node N` instead of the
reference's `mpirun -np N`; particles are sharded with MPI_start_end, class sums go through one
RCCL all-reduce.  Outputs: per iteration `aqm%03d.<ext>` class averages (reference :519,564), at the end
`params.txt` rows `idx angle_psi shift_x shift_y mirror class` / `initial2Dparams.txt` rows
`alpha sx sy mirror`.  --function=ref_ali2d (the default) runs the FSC-fitted tangent filter and the
centring on the device; any other name is rejected.  An optional mask file replaces model_circle(ou); with an
.hdf input stack the parameters also go into headers (EMAN.xform.align2d / assign / ID): of a copy of the stack under
outdir by default, of the input stack itself with --header_writeback (the reference's behaviour).
Options that would change the result and are not implemented (--CTF, --center 2..5, --random_method, --Fourvar,
--dst != 0, --mode other than F, --randomize, --orient) end the run with an error; --MPI / --EQ select paths with the same results and are only noted.
"""
import argparse
import os
import sys

import numpy as np


def _common(p):
    p.add_argument("--ir", type=float, default=1)
    p.add_argument("--ou", type=float, default=-1)
    p.add_argument("--rs", type=float, default=1)
    p.add_argument("--ts", default="1")
    p.add_argument("--maxit", type=float, default=10)
    p.add_argument("--CTF", action="store_true")
    p.add_argument("--snr", type=float, default=1.0)
    p.add_argument("--function", default="ref_ali2d")
    p.add_argument("--rand_seed", type=int, default=1000)
    p.add_argument("--gpu_devices", default="")
    p.add_argument("--gpu_info", action="store_true")
    p.add_argument("--MPI", action="store_true")
    p.add_argument("--EQ", action="store_true")
    p.add_argument("--ext", default="hdf", help="format of the written stacks: hdf (EMAN2 MDF, as the reference) | mrcs | npy")
    p.add_argument("--header_writeback", action="store_true",
                   help="write xform.align2d / assign / ID into the headers of the HDF INPUT stack itself, as the reference does "
                        "(the stack is rebuilt and replaced; refused when it holds attributes this writer cannot carry over). "
                        "Default: the stack with the new headers is written to <outdir>/<stack name> instead")
    p.add_argument("--no_header_writeback", action="store_true", help="write no stack with xform.align2d / assign / ID headers at all")


def _first(v):
    """the reference uses only stage 0 of "4 2 1 1"-style lists (test_reffree_gpu_align.py:355-357)"""
    return float(str(v).split()[0])


def _user_func(name):
    """--function: the reference resolves the name in sp_user_functions (user_functions.factory[name], test_mref_gpu_align.py:555);
    the engine implements its default, ref_ali2d, on the device.  The extension point itself is kept: "--function=file.py:name"
    (or "package.module:name") loads a Python callable  name(refs, buf, counts) -> refs  that receives the new class averages
    as a CUDA tensor [R][nx][nx], the reduced ClassSumBuffer and the class sizes, and returns the references of the next
    iteration (they are re-normalised under the mask afterwards, as at :563)."""
    if name in ("ref_ali2d", "none", "None", ""):
        return "ref_ali2d" if name == "ref_ali2d" else None
    if ":" in name:
        import importlib
        import importlib.util
        mod_name, func = name.rsplit(":", 1)
        if mod_name.endswith(".py") or os.path.sep in mod_name:
            spec = importlib.util.spec_from_file_location("ralign_user_function", mod_name)
            if spec is None:
                raise SystemExit("--function: cannot load %s" % mod_name)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
        else:
            mod = importlib.import_module(mod_name)
        if not callable(getattr(mod, func, None)):
            raise SystemExit("--function: %s has no callable %s" % (mod_name, func))
        return getattr(mod, func)
    raise SystemExit("--function=%s is not implemented by the MI355X engine (ref_ali2d | none | file.py:callable)" % name)


def _reject_unimplemented(args, reffree):
    """options of the reference's command lines that change the RESULT and that the engine does not implement are an error
    (exit code 2), not a silent no-op: --center 2..5 (only -1 | 0 | 1: average-centre rule / none / phase_cog),
    --random_method SHC | SCF, --Fourvar, --mode other than F (full rings), --CTF, --dst != 0 (the angular restriction
    `delta` of the CPU twin's ali2d_single_iter -> Util.Crosrng_ms_delta, whose source is not in the reference tree; the
    reference's GPU driver itself keeps delta = 0.0, test_reffree_gpu_align.py:307)."""
    bad = []
    if int(args.center) not in (-1, 0, 1):
        bad.append("--center %g (implemented: -1, 0, 1)" % args.center)
    if args.CTF:
        bad.append("--CTF")
    if reffree:
        if args.random_method not in ("", "none", "None"):
            bad.append("--random_method %s" % args.random_method)
        if args.Fourvar:
            bad.append("--Fourvar")
        if str(args.mode).upper() != "F":
            bad.append("--mode %s (implemented: F, full rings)" % args.mode)
        if args.dst != 0:
            bad.append("--dst %g (angular restriction, Crosrng_ms_delta)" % args.dst)
        # sub-options of the "friedel" mode (test_reffree_gpu_align.py:933-934): accepted like the reference's parser does,
        # an error when set -- the reference's driver never reads them, the SPHIRE code they belong to is not on this path
        if args.randomize:
            bad.append("--randomize")
        if args.orient:
            bad.append("--orient")
    if bad:
        raise SystemExit("not implemented by the MI355X engine: " + "; ".join(bad))


def _write_headers(mdfio, args, params, assign=None, ids=None):
    """xform.align2d / assign / ID headers: into a copy of the stack under outdir (default) or, with --header_writeback,
    into the input stack itself (refused, with the reason, when that would lose header items)"""
    dst = args.stack if args.header_writeback else os.path.join(args.outdir, os.path.basename(args.stack))
    if not args.header_writeback and os.path.abspath(dst) == os.path.abspath(args.stack):
        # the input stack lives in outdir: the copy must not replace it behind the user's back
        stem, ext = os.path.splitext(os.path.basename(args.stack))
        dst = os.path.join(args.outdir, stem + "_aligned" + ext)
    lost = mdfio.write_alignment_headers(args.stack, dst, params, assign=assign, ids=ids)
    if lost:
        print("warning: %s does not carry the header items %s of %s (types this writer cannot encode)" % (dst, ", ".join(lost), args.stack),
              file=sys.stderr)


def _setup(args):
    import torch
    from . import dist as rdist
    if args.gpu_devices:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", args.gpu_devices)     # CUDA_VISIBLE_DEVICES analogue (:1239-1250)
    rank, local, world = rdist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: the alignment engine has no CPU path")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if args.gpu_info:
        from . import api
        api.load_library().print_gpu_info(local)
        raise SystemExit(0)
    # flags that select another code path of the reference with the same results: reported, not an error
    for flag, name in ((args.MPI, "--MPI (the reference's CPU path; this build always runs the GPU engine)"),
                       (args.EQ, "--EQ (equal-size classes: the reference's GPU path ignores it as well)")):
        if flag and rank == 0:
            print("note: %s" % name, file=sys.stderr)
    return rank, local, world


def main_mref(argv=None):
    p = argparse.ArgumentParser(prog="test_mref_gpu_align.py")
    p.add_argument("stack"); p.add_argument("refstack"); p.add_argument("outdir"); p.add_argument("maskfile", nargs="?")
    p.add_argument("--xr", default="0"); p.add_argument("--yr", default="0")
    p.add_argument("--center", type=float, default=1)
    _common(p)
    args = p.parse_args(argv)
    _reject_unimplemented(args, reffree=False)
    rank, local, world = _setup(args)
    from . import stackio, dist as rdist
    from .mref import MrefAligner
    total = stackio.stack_size(args.stack)
    lo, hi = rdist.shard_range(total, world, rank)
    data = stackio.read_stack(args.stack, lo, hi)                # every rank reads its own slice only (:1358-1375)
    refs = stackio.read_stack(args.refstack)
    nx = data.shape[-1]
    ou = int(args.ou) if args.ou > 0 else nx // 2 - 2            # last_ring default (:311)
    mask = stackio.read_stack(args.maskfile)[0] if args.maskfile else None      # get_image(maskfile) (:317-319)
    xr, yr, ts = _first(args.xr), _first(args.yr), _first(args.ts)
    al = MrefAligner(data, refs, ou, xr, yr, ts, int(args.ir), int(args.rs), device=local, index0=lo,
                     total_nima=total, rand_seed=args.rand_seed, preprocess=True, mask=mask)
    if rank == 0:
        os.makedirs(args.outdir, exist_ok=True)
    maxit = int(args.maxit) if int(args.maxit) > 0 else 10
    ufunc = _user_func(args.function)
    for it in range(maxit):
        counts = al.iterate(ufunc, int(args.center))
        # members of every class (global particle numbers), as the reference gathers them every iteration (:504-515)
        ids = al.params()["ref_id"].astype(np.int32)
        if world > 1:
            import torch.distributed as td
            parts_ids = [None] * world
            td.all_gather_object(parts_ids, ids)
            ids = np.concatenate(parts_ids)
        if rank == 0:
            members = [np.nonzero(ids == j)[0].astype(np.int32) for j in range(al.nref)]
            if args.ext in ("hdf", "h5"):
                from . import mdfio
                mdfio.write_mdf_stack(os.path.join(args.outdir, "aqm%03d.%s" % (it, args.ext)), al.refs.cpu().numpy(),
                                      [{"members": m if m.size else np.array([-1], np.int32), "n_objects": np.int32(m.size)} for m in members])
            else:
                stackio.write_stack(os.path.join(args.outdir, "aqm%03d.%s" % (it, args.ext)), al.refs.cpu().numpy())
            if al.class_fsc_curves is not None:
                # fsc(refi[j][0], refi[j][1], 1.0, "drm%03d%04d.txt") of every class that did not vanish (:533)
                fsc_j, npt_j = al.class_fsc_curves
                for j in range(al.nref):
                    if counts[j] >= 4:
                        n_sh = fsc_j.shape[1]
                        stackio.write_text_rows(os.path.join(args.outdir, "drm%03d%04d.txt" % (it, j)),
                                                [(i / (2.0 * (n_sh - 1)), float(fsc_j[j, i]), float(npt_j[j, i])) for i in range(n_sh)])
            if al.filter_params:
                print("Tangent filter:  cut-off frequency = %10.3f        fall-off = %10.3f" % al.filter_params[-1])
            print("ITERATION #%3d" % (it + 1))
            for j, c in enumerate(counts):
                print("   group #%3d   number of particles = %7d" % (j, c))
    r = al.params()
    rows = [(lo + i, float(r["alpha"][i]), float(r["sx"][i]), float(r["sy"][i]), int(r["mirror"][i]), int(r["ref_id"][i]))
            for i in range(hi - lo)]
    if world > 1:
        import torch.distributed as td
        gathered = [None] * world
        td.all_gather_object(gathered, rows)
        rows = [x for part in gathered for x in part]
    if rank == 0:
        stackio.write_text_rows(os.path.join(args.outdir, "params.txt"), rows)
        stackio.write_stack(os.path.join(args.outdir, "multi_ref.%s" % args.ext), al.refs.cpu().numpy())
        if os.path.splitext(args.stack)[1].lower() in (".hdf", ".h5") and not args.no_header_writeback:
            # xform.align2d / assign / ID into the stack's headers (set_params2D :588; output_attr / write_attr of
            # test_mref_cheng_yu_bdb_cuda.py:114-203)
            from . import mdfio
            rows.sort(key=lambda x: x[0])
            _write_headers(mdfio, args, [x[1:5] for x in rows], assign=[x[5] for x in rows], ids=[x[0] for x in rows])
    al.close()
    return 0


def main_reffree(argv=None):
    p = argparse.ArgumentParser(prog="test_reffree_gpu_align.py")
    p.add_argument("stack"); p.add_argument("outdir"); p.add_argument("maskfile", nargs="?")
    p.add_argument("--xr", default="4 2 1 1"); p.add_argument("--yr", default="-1")
    p.add_argument("--center", type=float, default=-1)
    p.add_argument("--nomirror", action="store_true"); p.add_argument("--dst", type=float, default=0.0)
    p.add_argument("--Fourvar", action="store_true"); p.add_argument("--mode", default="F")
    p.add_argument("--random_method", default="")
    p.add_argument("--randomize", action="store_true"); p.add_argument("--orient", action="store_true")
    p.add_argument("--all_stages", action="store_true", help="run every stage of --xr/--ts lists (SPHIRE ali2d_base); the reference's "
                   "GPU driver runs stage 0 only, which is the default here")
    p.add_argument("--auto_stop", action="store_true", help="with --maxit 0: stop a stage when the criterion decreases (the rule the "
                   "reference's comments state; its driver computes the flag and never tests it, which is the default here)")
    _common(p)
    args = p.parse_args(argv)
    _reject_unimplemented(args, reffree=True)
    rank, local, world = _setup(args)
    from . import stackio, dist as rdist
    from .mref import RefFreeAligner
    total = stackio.stack_size(args.stack)
    lo, hi = rdist.shard_range(total, world, rank)
    data = stackio.read_stack(args.stack, lo, hi)
    nx = data.shape[-1]
    ou = int(args.ou) if args.ou > 0 else nx // 2 - 2
    mask = stackio.read_stack(args.maskfile)[0] if args.maskfile else None
    # "--xr '4 2 1 1' --ts '2 1 0.5 0.25'": one stage per entry, --maxit iterations each; --maxit 0 = 10 with auto-stop
    al = RefFreeAligner(data, ou, args.xr, args.yr, args.ts, int(args.ir), int(args.rs), device=local, index0=lo,
                        total_nima=total, nomirror=args.nomirror, mask=mask)
    max_iter = 10 if int(args.maxit) == 0 else int(args.maxit)
    auto_stop = args.auto_stop and int(args.maxit) == 0
    al.track_pixel_error = True
    if rank == 0:
        os.makedirs(args.outdir, exist_ok=True)
    ufunc = _user_func(args.function)
    if callable(ufunc):
        raise SystemExit("--function: a Python callable is supported by test_mref_gpu_align.py only (ref_ali2d | none here)")
    a0, it = -1.0e22, 0
    aqc, aqf = [], []
    for n_step in range(len(al.stages) if args.all_stages else 1):
        al.set_stage(n_step)
        for _ in range(max_iter):
            it += 1
            a1 = al.iterate(int(args.center), ufunc)
            if rank == 0:
                print("Iteration #%4d   X range = %5.2f   Y range = %5.2f   Step = %5.2f   Criterion = %15.8e" % ((it,) + al.stages[n_step] + (a1,)))
                # aqc / aqf: one stack each, image number = iteration (tavg.write_image(".../aqc.hdf", total_iter - 1), :383, :420;
                # aqc = the average REDUCED OVER THE RANKS before the user function, (ave1 + ave2) / total_nima, of every iteration
                # including the raw sum_oe average of the first; aqf = the filtered / centred average the particles are aligned to)
                aqc.append(al.raw_avg.cpu().numpy().copy())
                stackio.write_stack(os.path.join(args.outdir, "aqc.%s" % args.ext), np.stack(aqc))
                aqf.append(al.tavg[0].cpu().numpy().copy())
                stackio.write_stack(os.path.join(args.outdir, "aqf.%s" % args.ext), np.stack(aqf))
                if al.pixel_errors:
                    mc, pe = al.pixel_errors[-1]
                    print("  mirror consistent = %d of %d (this rank), mean squared pixel error = %.4f" % (mc, al.n, pe / max(mc, 1)))
            if a1 < a0:
                if auto_stop:
                    break
            else:
                a0 = a1
    r = al.params()
    rows = [(float(r["alpha"][i]), float(r["sx"][i]), float(r["sy"][i]), int(r["mirror"][i])) for i in range(hi - lo)]
    if world > 1:
        import torch.distributed as td
        gathered = [None] * world
        td.all_gather_object(gathered, rows)
        rows = [x for part in gathered for x in part]
    if rank == 0:
        os.makedirs(args.outdir, exist_ok=True)
        stackio.write_text_rows(os.path.join(args.outdir, "initial2Dparams.txt"), rows)
        stackio.write_stack(os.path.join(args.outdir, "aqfinal.%s" % args.ext), al.tavg.cpu().numpy())
        if os.path.splitext(args.stack)[1].lower() in (".hdf", ".h5") and not args.no_header_writeback:
            from . import mdfio        # set_params2D(img, [angle, shift_x, shift_y, mirror, 1.0], "xform.align2d") (test_reffree.py:453)
            _write_headers(mdfio, args, rows)
    al.close()
    return 0
