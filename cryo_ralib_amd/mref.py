"""Host-side iteration drivers of the alignment path.

`MrefAligner` / `mref_ali2d_gpu` mirror the reference's `mref_ali2d_gpu`
(test_mref_gpu_align.py:222-612): preprocessing (:333-345), one search + transform +
class accumulation per iteration (:408-484), the cross-rank reduction (:495-499), the
vanished-class rule (:523-528), (even+odd)/n and re-normalisation (:534-535, 563), and the
AlignParam -> xform.align2d conversion (:578-588).  `RefFreeAligner` / `ali2d_base_gpu`
mirror `ali2d_base_gpu_isac_CLEAN` (test_reffree_gpu_align.py:153-577).

All heavy work is in the HIP engine (api.Engine); torch supplies device memory, the stream
and the RCCL collective.  `user_func="ref_ali2d"` runs the reference's default user function
(--function=ref_ali2d, :1153: FSC-fitted tangent low-pass + centring, :531-563) on the device
(SURVEY.md §8 f-1); `user_func=None` is an identity user function (what bench.py times).
"""
import random

import numpy as np
import torch

from . import api, dist, geometry


class _Pending:
    """a device value on its way to the host: copied into pinned memory on the current stream, with an event behind the copy.
    get() waits for THAT event only (hipEventSynchronize, not a stream synchronisation): kernels queued after the copy keep the
    GPU busy while the host reads the value."""

    def __init__(self, t):
        self.host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        self.host.copy_(t, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(t.device))

    def get(self):
        self.event.synchronize()
        return self.host.numpy()


def _fit_params(a):
    """(fl, aa) of the device fit as Python floats, with ref_ali2d's clamps in double like the host code they replace
    (the device applied them in float32: 0.2f is 0.20000000298)"""
    return max(min(0.4, float(a[0])), 0.12), min(float(a[1]), 0.2)


class _Lagged:
    """per-iteration bookkeeping lists (class sizes, filter parameters, FSC curves, centres) whose entries are read back from
    the device only when somebody looks at them"""

    def __init__(self):
        self._done, self._todo = [], []

    def push(self, pending, convert):
        self._todo.append((pending, convert))

    def append(self, value):
        self.resolve()
        self._done.append(value)

    def resolve(self):
        for pending, convert in self._todo:
            self._done.append(convert(pending.get()))
        self._todo = []
        return self._done


class MrefAligner:
    def __init__(self, particles, refs, ou, xr, yr, ts=1.0, ir=1, rs=1, device=0, index0=0, total_nima=None,
                 rand_seed=1000, preprocess=True, chunk=0, myid=0, main_node=0, mask=None, state_roundtrip=True, refine=None):
        """particles: [n][nx][nx] float32 numpy array or CUDA tensor holding THIS rank's shard;
        refs: [R][nx][nx]; index0 = global index of particles[0] (even/odd split).
        state_roundtrip: rebuild the shift every search starts from out of the float32 (alpha, sx, sy) of the previous
        iteration with inverse_transform2, as the reference's loop does (test_mref_gpu_align.py:1024-1026); False carries
        the exact accumulated shift instead (algebraically the same; differs by rounding at edge-limited windows).
        refine: threshold of the sub-bin angle refinement (api.Engine.set_refine; None = the engine's default, -1 = every
        particle: alpha / sx / sy then equal the CPU path's to the last bit wherever the integer winner agrees)."""
        self.state_roundtrip = bool(state_roundtrip)
        self._have_params = False
        self.dev = torch.device("cuda", device)
        self.particles = self._to_dev(particles)
        self.refs = self._to_dev(refs).clone()
        self.n, self.nx = self.particles.shape[0], self.particles.shape[-1]
        self.nref = self.refs.shape[0]
        self.ou, self.xr, self.yr, self.ts = int(ou), float(xr), float(yr), float(ts)
        self.index0 = int(index0)
        self.total_nima = int(total_nima if total_nima is not None else self.n)
        self.myid, self.main_node = myid, main_node
        # "Shift or radius is too large - particle crosses image boundary" (:314-315) is raised by ra_create
        self.engine = api.Engine(self.nx, self.ou, self.xr, self.yr, self.ts, self.nref, api.RA_MODE_MREF,
                                 first_ring=ir, ring_skip=rs, device=device, chunk=chunk)
        self.engine.use_current_stream()
        if refine is not None:
            self.engine.set_refine(refine)
        if mask is None:        # "mask = model_circle(last_ring, nx, nx)" unless a mask file was given (:317-321)
            self.mask = torch.from_numpy(geometry.model_circle(self.ou, self.nx, self.nx)).to(self.dev)
        else:
            self.mask = self._to_dev(mask).reshape(self.nx, self.nx)
            self.engine.set_mask(self.mask)
        self.state = self.engine.new_state(self.n)
        self.result = self.engine.new_result(self.n)
        self.buf = dist.ClassSumBuffer(self.nref, self.nx, self.dev)
        self.iteration = 0
        self.rng = random.Random(rand_seed)       # seed(rand_seed) on the main node (:352)
        self._class_sizes = _Lagged()
        self._filter_params, self._fsc_curves, self._centres = _Lagged(), _Lagged(), _Lagged()     # per iteration, user_func="ref_ali2d"
        self._have_class_fsc = False
        # device-side results of the reference update (ra_class_fsc_fit / ra_filter_references_dev): nothing of it comes back to
        # the host inside an iteration
        self._fit = torch.zeros(8, device=self.dev)
        self._curve = torch.zeros(3 * self.engine.fsc_len, device=self.dev)
        self._cs_out = torch.zeros((self.nref, 2), device=self.dev)
        if preprocess:
            self._normalize_refs_all()
            self.engine.normalize_particles(self.particles)    # :342

    def _to_dev(self, a):
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(np.ascontiguousarray(a, np.float32))
        return a.to(self.dev, dtype=torch.float32).contiguous()

    def _normalize_refs(self, idx=None):
        # normalize.mask(no_sigma=1) (:336, :563): mean 0, sigma 1 under the mask
        sel = self.mask > 0.5
        idx = list(range(self.nref)) if idx is None else list(idx)
        if not idx:
            return
        sub = self.refs[idx]
        v = sub[:, sel].double()
        n = v.shape[1]
        mean = (v.sum(1) / n).float()
        var = ((v * v).sum(1) - v.sum(1) ** 2 / n) / (n - 1)
        sigma = var.float().sqrt()
        self.refs[idx] = (sub - mean[:, None, None]) / sigma[:, None, None]

    def _normalize_refs_all(self):
        self._normalize_refs(None)

    def search(self):
        """pre_align_fetch("ref_batch") + mref_align_run + kernel_sum_oe of one iteration (:410-453)."""
        self.engine.set_references(self.refs)
        if self.state_roundtrip and self._have_params:
            self.engine.state_from_params(self.result, self.state)         # alphai, sxi, syi = inverse_transform2(alpha, sx, sy)
        self.engine.align(self.particles, self.state, self.result)
        self._have_params = True
        self.buf.zero_()
        self.engine.transform_accumulate(self.particles, self.result, self.index0, None, self.buf.sums,
                                         self.buf.counts_i)

    # per-iteration records; read back lazily (the values are device results of iterations that may still be running)
    @property
    def class_sizes(self):
        return self._class_sizes.resolve()

    @property
    def filter_params(self):
        return self._filter_params.resolve()

    @property
    def fsc_curves(self):
        return self._fsc_curves.resolve()

    @property
    def centres(self):
        return self._centres.resolve()

    @property
    def class_fsc_curves(self):
        """per class: what the reference writes to drm%03d%04d.txt (:533) for the last iteration (fetched on demand)"""
        return self.engine.last_class_fsc() if self._have_class_fsc else None

    def _reseed(self, vanished):
        for j in vanished:
            # "if vanished, put a random image (only from main node!) there" (:523-528)
            k = self.rng.randint(0, self.n - 1)
            img = self.particles[min(k, self.n - 1)].clone()       # (other ranks: any image of the right shape, overwritten)
            dist.broadcast(img, src=self.main_node)
            self.refs[j].copy_(img)

    def reduce_and_update(self, user_func=None, center=1):
        """cross-rank sum (:495-499) then the reference update every rank repeats (:517-575).

        No stream synchronisation: the class sizes (vanished-class rule: the reference draws from its RNG only when a class
        vanished, so the host has to see them) travel to pinned memory behind an event, and the host looks at them after it has
        queued the kernels that do not depend on them; with user_func="ref_ali2d" the FSC, fit_tanh, its clamps, the filter and
        the centring stay on the device (ra_class_fsc_fit / ra_filter_references_dev)."""
        self.buf.all_reduce()
        pc = _Pending(self.buf.counts_i)
        if user_func == "ref_ali2d":
            # the default user function on the device: fsc per class (:531), averaged (:537-548), sp_user_functions.ref_ali2d =
            # fit_tanh + clamps + filt_tanl + center_2D(center), normalize.mask (:563)
            self.engine.class_fsc_fit(self.buf.sums, self.buf.counts_i, self._fit, self._curve, 4, masked=False)
            self.engine.class_averages(self.buf.sums, self.buf.counts_i, self.refs, 4)
            self._have_class_fsc = True
        else:
            self.engine.update_references(self.buf.sums, self.buf.counts_i, self.refs, 4)
        counts = pc.get().copy()
        self._class_sizes.append(counts.copy())
        vanished = [j for j in range(self.nref) if counts[j] < 4]
        if user_func == "ref_ali2d":
            if len(vanished) == self.nref:
                raise api.EngineError("every class vanished: no FSC to fit the filter to")
            self._reseed(vanished)
            self.engine.filter_references_dev(self.refs, self._fit, center=1 if center == 1 else 0, normalize=True, cs_out=self._cs_out)
            n = self.engine.fsc_len
            self._filter_params.push(_Pending(self._fit[:2]), _fit_params)
            self._fsc_curves.push(_Pending(self._curve), lambda a: [list(map(float, a[:n])), list(map(float, a[n:2 * n])), list(map(float, a[2 * n:3 * n]))])
            self._centres.push(_Pending(self._cs_out), lambda a: a.copy())
            self.iteration += 1
            return counts
        self._reseed(vanished)
        if user_func is not None:
            self.refs = user_func(self.refs, self.buf, counts)
        if user_func is not None:
            self._normalize_refs_all()      # :563
        elif vanished:
            self._normalize_refs(vanished)  # the others were normalised by the update kernel
        self.iteration += 1
        return counts

    def iterate(self, user_func=None, center=1):
        self.search()
        return self.reduce_and_update(user_func, center)

    def params(self):
        """per-particle (alpha, sx, sy, mirror, ref_id, peak) of the last search."""
        r = api.Engine.result_to_numpy(self.result)
        return r

    def align_params(self):
        """AlignParam view (shift_x, shift_y, angle, mirror, ref_id) as the reference's library
        exposes it, and the EMAN2 conversion of :578-588."""
        r = self.params()
        st = self.state.cpu().numpy()
        return geometry.alignparam_to_eman2(r["alpha"], st[:, 0], st[:, 1], r["mirror"]), r["ref_id"]

    def close(self):
        self.engine.close()


def mref_ali2d_gpu(stack, refim, ou, xrng, yrng, step=1.0, ir=1, rs=1, maxit=10, rand_seed=1000, device=0,
                   index0=0, total_nima=None, user_func=None, chunk=0, on_iteration=None, center=1):
    """Multi-reference alignment of this rank's shard `stack` against `refim`
    (mirror of mref_ali2d_gpu, test_mref_gpu_align.py:222).  Returns
    (params records, class averages [R][nx][nx] numpy, list of class-size arrays)."""
    al = MrefAligner(stack, refim, ou, xrng, yrng, step, ir, rs, device, index0, total_nima, rand_seed, True, chunk)
    max_iter = int(maxit) if int(maxit) > 0 else 10
    for it in range(max_iter):
        counts = al.iterate(user_func, center)
        if on_iteration is not None:
            on_iteration(it, al, counts)
    al.engine.sync()
    out = al.params().copy(), al.refs.cpu().numpy(), al.class_sizes
    al.close()
    return out


def _as_list(v):
    """"4 2 1 1" / [4, 2, 1, 1] / 4 -> list of floats (get_input_from_string, test_reffree_gpu_align.py:215-216)"""
    if isinstance(v, str):
        return [float(t) for t in v.replace(",", " ").split()]
    try:
        return [float(t) for t in v]
    except TypeError:
        return [float(v)]


class RefFreeAligner:
    """single-reference alignment to the running average (ali2d_base_gpu_isac_CLEAN,
    test_reffree_gpu_align.py:153-577; CPU twin ali2d_base -> ali2d_single_iter -> ormq).

    xr / yr / ts may be lists ("--xr '4 2 1 1' --ts '2 1 0.5 0.25'"): one search window per stage
    (test_reffree_gpu_align.py:215-216); `set_stage(i)` switches the engine to stage i (reset_shifts, :355-357).
    The engine is sized once for the stage with the most search offsets and the widest range."""

    def __init__(self, particles, ou, xr, yr, ts=1.0, ir=1, rs=1, device=0, index0=0, total_nima=None,
                 preprocess=False, chunk=0, nomirror=False, mask=None, refine=None):
        self.dev = torch.device("cuda", device)
        if isinstance(particles, np.ndarray):
            particles = torch.from_numpy(np.ascontiguousarray(particles, np.float32))
        self.particles = particles.to(self.dev, dtype=torch.float32).contiguous()
        self.n, self.nx = self.particles.shape[0], self.particles.shape[-1]
        self.ou = int(ou)
        self.index0 = int(index0)
        self.total_nima = int(total_nima if total_nima is not None else self.n)
        xr, ts = _as_list(xr), _as_list(ts)
        yr = _as_list(yr)
        if len(yr) == 1 and yr[0] < 0:
            yr = list(xr)                                   # "--yr -1": same as xr
        nst = max(len(xr), len(ts), len(yr))
        pad = lambda l: l + [l[-1]] * (nst - len(l))
        self.stages = list(zip(pad(xr), pad(yr), pad(ts)))
        # capacity window: the widest range, and a step that yields the largest offset count of any stage
        rmax = max(max(x, y) for x, y, _ in self.stages)
        nk = max(max(int(x / t), int(y / t)) for x, y, t in self.stages)
        cap_step = (rmax / nk) * (1.0 - 1e-6) if nk > 0 else 1.0
        self.engine = api.Engine(self.nx, self.ou, rmax, rmax, cap_step, 1, api.RA_MODE_REFFREE, first_ring=ir, ring_skip=rs,
                                 device=device, chunk=chunk)
        self.engine.use_current_stream()
        if refine is not None:
            self.engine.set_refine(refine)
        self.stage = -1
        self.set_stage(0)
        if nomirror:
            self.engine.set_nomirror(True)
        if mask is None:
            self.mask = torch.from_numpy(geometry.model_circle(self.ou, self.nx, self.nx)).to(self.dev)
        else:
            if isinstance(mask, np.ndarray):
                mask = torch.from_numpy(np.ascontiguousarray(mask, np.float32))
            self.mask = mask.to(self.dev, dtype=torch.float32).reshape(self.nx, self.nx).contiguous()
            self.engine.set_mask(self.mask)
        if preprocess:
            self.engine.normalize_particles(self.particles)
        self.state = self.engine.new_state(self.n)
        self.result = self.engine.new_result(self.n)
        self.buf = dist.ClassSumBuffer(1, self.nx, self.dev, extra=2)
        self.tavg = torch.zeros((1, self.nx, self.nx), device=self.dev)
        self.raw_avg = None
        self._cs_dev = torch.zeros((1, 2), device=self.dev)      # centre correction of the current iteration (device)
        self._cs_pending = None
        self._fit = torch.zeros(8, device=self.dev)
        self._curve = torch.zeros(3 * self.engine.fsc_len, device=self.dev)
        self.iteration = 0
        self._criteria, self._filter_params = _Lagged(), _Lagged()
        self.track_pixel_error = False      # the drivers' per-iteration bookkeeping (mirror-consistent count, summed pixel error)
        self.pixel_errors = []

    def set_stage(self, i):
        """search window of stage i: cu_module.reset_shifts(xrng[N_step], step[N_step]) (test_reffree_gpu_align.py:357)"""
        if i != self.stage:
            x, y, t = self.stages[i]
            self.engine.reset_shifts(x, y, t)
            self.stage = i

    def _sum_oe_raw(self):
        # iteration 0: plain even/odd sums of the raw particles (sum_oe, :365)
        self.buf.zero_()
        par = (torch.arange(self.n, device=self.dev) + self.index0) % 2
        self.buf.sums[0, 0] = self.particles[par == 0].sum(0)
        self.buf.sums[0, 1] = self.particles[par == 1].sum(0)
        self.buf.counts_i[0] = self.n

    @property
    def criteria(self):
        return self._criteria.resolve()

    @property
    def filter_params(self):
        return self._filter_params.resolve()

    @property
    def cs(self):
        """centre correction (cs[0], cs[1]) applied in the last iteration"""
        if self._cs_pending is None:
            return [0.0, 0.0]
        a = self._cs_pending.get()
        return [float(a[0, 0]), float(a[0, 1])]

    def iterate(self, center=0, user_func=None):
        """one iteration of ali2d_base_gpu_isac_CLEAN (:361-540).  user_func="ref_ali2d": fsc_mask (:384), fit_tanh and its
        clamps, the tangent filter, and the centring of the average (fshift by the average centre for center=-1, :403-410;
        center_2D method for center > 0) on the device; None keeps the raw average.

        Nothing the iteration computes passes through the host: the criterion, the centre and the filter parameters are device
        values (read back behind events, after the iteration's kernels are queued), so the stream is never synchronised."""
        if self.iteration == 0:
            self._sum_oe_raw()
        self.buf.all_reduce()
        # tavg = (ave1 + ave2) / total_nima (:380); criterion a1 = sum_mask tavg^2 (:396)
        self.tavg[0] = (self.buf.sums[0, 0] + self.buf.sums[0, 1]) / float(self.total_nima)
        # the average as reduced over the ranks, before the user function and the centring: what the reference writes to
        # aqc.hdf at image total_iter - 1 in every iteration, iteration 0 (sum_oe of the raw stack) included (:365-383)
        self.raw_avg = self.tavg[0].clone()
        # (no boolean-mask indexing here: it reads the number of selected elements back and so synchronises the host with the
        # stream at the top of every iteration -- the GPU then idles while the host queues the kernels in front of the search)
        pa1 = _Pending(torch.where(self.mask > 0.5, self.tavg[0], torch.zeros_like(self.tavg[0])).double().square().sum().float().reshape(1))
        self._cs_dev.zero_()
        if center == -1 and self.iteration > 0:
            # average-centre rule cs = (sum +-sx, sum sy) / N (:403-410) in double, rounded to float32 once like the host value
            # it replaces; with user_func the average is shifted by -cs on the device (fshift), the particle parameters are
            # corrected inside ra_state_from_params
            self._cs_dev[0] = (self.buf.extra_f[:2].double() / float(self.total_nima)).float()
        if user_func == "ref_ali2d":
            self.engine.class_fsc_fit(self.buf.sums, self.buf.counts_i, self._fit, self._curve, 1, masked=True)
            self._filter_params.push(_Pending(self._fit[:2]), _fit_params)
            if center == -1:
                self.engine.filter_references_dev(self.tavg, self._fit, center=-1, cs_in=self._cs_dev, normalize=False)
            else:
                self.engine.filter_references_dev(self.tavg, self._fit, center=1 if center == 1 else 0, normalize=False, cs_out=self._cs_dev)
        elif center == -1 and self.iteration > 0:
            # no user function: the average is still shifted by -cs, together with the parameter correction
            # (fshift(tavg, -cs[0], -cs[1]), test_reffree_gpu_align.py:403-410)
            self.engine.filter_references_dev(self.tavg, None, center=-1, cs_in=self._cs_dev, normalize=False)
        self._cs_pending = _Pending(self._cs_dev)
        old = None
        if self.track_pixel_error and self.iteration > 0:
            old = self.result.clone()                       # old_ali_params (test_reffree_gpu_align.py:833-838)
        self.engine.set_references(self.tavg)
        # ali2d_single_iter: combine_params2(alpha, sx, sy, mirror, 0, -cs[0], -cs[1], 0), inverse_transform2 -> sxi, syi
        # (iteration 0: the header parameters are zero, the centre of the first average is not)
        self.engine.state_from_params_dev(self.result, self.state, self._cs_dev)
        self.engine.align(self.particles, self.state, self.result, None)
        self.buf.zero_()
        self.engine.transform_accumulate(self.particles, self.result, self.index0, None, self.buf.sums,
                                         self.buf.counts_i)
        r = self.result.view(torch.float32)
        mir = self.result[:, 3]
        sx = r[:, 1].double()
        self.buf.extra_f[0] = torch.where(mir == 0, sx, -sx).sum().float()
        self.buf.extra_f[1] = r[:, 2].double().sum().float()
        if old is not None:
            # pixel_error / mirror_consistent of test_reffree_gpu_align.py:523-538 with sp_pixel_error.pixel_error_2D
            # (sin((a1 - a2) / 2) (2 r + 1))^2 + (sx1 - sx2)^2 + (sy1 - sy2)^2 (formula from SPHIRE, not in the reference tree)
            o = old.view(torch.float32)
            same = old[:, 3] == mir
            da = torch.deg2rad((o[:, 0] - r[:, 0]).double())
            err = (torch.sin(da / 2) * (2 * self.ou + 1)) ** 2 + (o[:, 1] - r[:, 1]).double() ** 2 + (o[:, 2] - r[:, 2]).double() ** 2
            self.pixel_errors.append((int(same.sum().item()), float(err[same].sum().item())))
        self.iteration += 1
        # the criterion was computed before the search was queued: reading it now costs the GPU nothing
        a1 = float(pa1.get()[0])
        self._criteria.append(a1)
        return a1

    def params(self):
        return api.Engine.result_to_numpy(self.result)

    def close(self):
        self.engine.close()


def ali2d_base_gpu(stack, ou, xrng, yrng, step=1.0, ir=1, rs=1, maxit=10, device=0, index0=0, total_nima=None,
                   center=0, chunk=0, user_func=None, nomirror=False, on_iteration=None, all_stages=False, auto_stop=False):
    """mirror of ali2d_base_gpu_isac_CLEAN; returns (params records, final average, criteria).
    Rows of initial2Dparams.txt are (alpha, sx, sy, mirror) (test_reffree_gpu_align.py:561-569).

    Defaults follow the reference's GPU driver to the letter: xrng / yrng / step may be lists, but only stage 0 runs
    (N_step = 0, test_reffree_gpu_align.py:355-357), and maxit = 0 means 10 iterations -- the driver computes and
    broadcasts the auto-stop flag `again` (:422-433) and never tests it.
    all_stages=True runs every stage, `maxit` iterations each (SPHIRE's ali2d_base schedule, "--xr '4 2 1 1' --ts '2 1
    0.5 0.25'"); auto_stop=True (with maxit = 0) ends a stage with the iteration whose average scored below the best so
    far -- the rule the reference's comments state ("a0 should increase; stop algorithm when it decreases", :392-396)."""
    al = RefFreeAligner(stack, ou, xrng, yrng, step, ir, rs, device, index0, total_nima, False, chunk, nomirror)
    max_iter = 10 if int(maxit) == 0 else int(maxit)
    auto_stop = bool(auto_stop) and int(maxit) == 0
    a0 = -1.0e22
    total_iter = 0
    for n_step in range(len(al.stages) if all_stages else 1):
        al.set_stage(n_step)
        for _ in range(max_iter):
            total_iter += 1
            a1 = al.iterate(center, user_func)
            if on_iteration is not None:
                on_iteration(total_iter, al, a1)
            if a1 < a0:
                if auto_stop:
                    break
            else:
                a0 = a1
    al.engine.sync()
    out = al.params().copy(), al.tavg[0].cpu().numpy(), al.criteria
    al.close()
    return out
