"""ctypes binding of libralign_hip.so (include/ralign.h).

Mirrors the reference's ctypes boundary (test_mref_gpu_align.py:91-149: cu_module,
AlignConfig, AlignParam, get_c_ptr_array) and adds the handle-based ra_* API that works
on device pointers (torch CUDA tensors supply the memory and the stream).

There is no CPU fallback: if the HIP library is missing or fails to load, importing the
engine raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RALIGN_LIB") or os.path.join(_HERE, "libralign_hip.so")   # RALIGN_LIB: another build of the same library (kernel experiments)

RA_MODE_MREF = 0
RA_MODE_REFFREE = 1
RA_INTERP_BILINEAR = 0      # Util::bilinear in alrl_ms (EMAN2 2.31; default)
RA_INTERP_QUADRI = 1        # Util::quadri (older releases): size-generic kernels


class AlignConfig(ctypes.Structure):
    # reference: test_mref_gpu_align.py:112-123 / cuda/gpu_aln_common.h:62-74
    _fields_ = [("sbj_num", ctypes.c_uint), ("ref_num", ctypes.c_uint), ("img_dim", ctypes.c_uint),
                ("ring_num", ctypes.c_uint), ("ring_len", ctypes.c_uint),
                ("shift_step", ctypes.c_float), ("shift_rng_x", ctypes.c_float), ("shift_rng_y", ctypes.c_float)]


class AlignParam(ctypes.Structure):
    # reference: test_mref_gpu_align.py:125-134 / cuda/gpu_aln_common.h:76-83
    _fields_ = [("sbj_id", ctypes.c_int), ("ref_id", ctypes.c_int), ("shift_x", ctypes.c_float),
                ("shift_y", ctypes.c_float), ("angle", ctypes.c_float), ("mirror", ctypes.c_bool)]

    def __str__(self):
        return "s_%d/r_%d::(%d,%d;%.2f)" % (self.sbj_id, self.ref_id, self.shift_x, self.shift_y, self.angle) \
            + ("[M]" if self.mirror else "")


class RaConfig(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int), ("first_ring", ctypes.c_int), ("last_ring", ctypes.c_int),
                ("ring_skip", ctypes.c_int), ("xrng", ctypes.c_float), ("yrng", ctypes.c_float),
                ("step", ctypes.c_float), ("nref", ctypes.c_int), ("mode", ctypes.c_int),
                ("device", ctypes.c_int), ("chunk", ctypes.c_int)]


class RaOptions(ctypes.Structure):
    # include/ralign.h: ra_options
    _fields_ = [("interp", ctypes.c_int), ("normalize_ring", ctypes.c_int)]


# ra_result as a numpy record (32 bytes)
RESULT_DTYPE = np.dtype([("alpha", np.float32), ("sx", np.float32), ("sy", np.float32), ("mirror", np.int32),
                         ("ref_id", np.int32), ("peak", np.float32), ("angle_bin", np.int32),
                         ("shift_idx", np.int32)])

float_ptr = ctypes.POINTER(ctypes.c_float)
aln_param_ptr = ctypes.POINTER(AlignParam)

# every symbol include/ralign.h declares
EXPORTED_SYMBOLS = [
    "print_gpu_info", "gpu_clear", "pre_align_init", "pre_align_size_check", "pre_align_fetch",
    "pre_align_run", "pre_align_run_m", "mref_align_run", "mref_align_run_m", "get_num_ref", "reset_shifts",
    "ref_free_alignment_2D_init", "ref_free_alignment_2D_size_check", "ref_free_alignment_2D",
    "ref_free_alignment_2D_filter_references", "ra_isac_get_references", "ra_legacy_bytes",
    "ra_last_error", "ra_create", "ra_destroy", "ra_set_stream", "ra_num_shifts", "ra_maxrin", "ra_lcirc", "ra_search_path", "ra_search_tiled", "ra_search_offsets_per_pass", "ra_set_nomirror", "ra_set_mask",
    "ra_reset_shifts", "ra_set_references", "ra_get_prepared_references", "ra_align", "ra_state_from_params", "ra_set_refine", "ra_set_class_references", "ra_align_classes",
    "ra_debug_spectra", "ra_transform_accumulate", "ra_update_references", "ra_normalize_particles", "ra_sync", "ra_kernel_time",
    "ra_fsc_len", "ra_class_fsc", "ra_last_class_fsc", "ra_fit_tanh", "ra_class_averages", "ra_filter_references",
    "ra_state_from_params_dev", "ra_class_fsc_fit", "ra_filter_references_dev", "ra_last_refine_count",
    "ra_create_ex", "ra_set_normalize_ring", "ra_get_options", "ra_search_skips_offsets",
]

_lib = None


class EngineError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen libralign_hip.so and declare prototypes.  Raises if the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # torch bundles its own HIP runtime (libamdhip64.so.7); it must be the one already mapped when
    # this library is opened, otherwise the process ends up with two runtimes and no device
    import torch  # noqa: F401
    if not os.path.exists(p):
        raise EngineError("HIP alignment library not built: %s (run `python -m cryo_ralib_amd.build`)" % p)
    L = ctypes.CDLL(p)
    vp = ctypes.c_void_p
    L.ra_last_error.restype = ctypes.c_char_p
    L.ra_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(RaConfig)]
    L.ra_create_ex.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(RaConfig), ctypes.POINTER(RaOptions)]
    L.ra_set_normalize_ring.argtypes = [vp, ctypes.c_int]
    L.ra_get_options.argtypes = [vp, ctypes.POINTER(RaOptions)]
    L.ra_destroy.argtypes = [vp]
    L.ra_destroy.restype = None
    L.ra_set_stream.argtypes = [vp, vp]
    L.ra_num_shifts.argtypes = [vp]
    L.ra_maxrin.argtypes = [vp]
    L.ra_lcirc.argtypes = [vp]
    L.ra_search_path.argtypes = [vp]
    L.ra_search_tiled.argtypes = [vp]
    L.ra_search_skips_offsets.argtypes = [vp]
    L.ra_search_offsets_per_pass.argtypes = [vp]
    L.ra_set_nomirror.argtypes = [vp, ctypes.c_int]
    L.ra_set_mask.argtypes = [vp, vp]
    L.ra_reset_shifts.argtypes = [vp, ctypes.c_float, ctypes.c_float, ctypes.c_float]
    L.ra_set_references.argtypes = [vp, vp]
    L.ra_get_prepared_references.argtypes = [vp, vp]
    L.ra_align.argtypes = [vp, vp, ctypes.c_int, vp, vp, float_ptr]
    L.ra_set_class_references.argtypes = [vp, vp, ctypes.c_int]
    L.ra_align_classes.argtypes = [vp, vp, ctypes.c_int, vp, vp, vp]
    L.ra_state_from_params.argtypes = [vp, vp, ctypes.c_int, float_ptr, vp]
    L.ra_set_refine.argtypes = [vp, ctypes.c_float]
    L.ra_last_refine_count.argtypes = [vp]
    L.ra_state_from_params_dev.argtypes = [vp, vp, ctypes.c_int, vp, vp]
    L.ra_class_fsc_fit.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp]
    L.ra_filter_references_dev.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_int, vp]
    L.ra_transform_accumulate.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
    L.ra_update_references.argtypes = [vp, vp, vp, ctypes.c_int, vp]
    L.ra_normalize_particles.argtypes = [vp, vp, ctypes.c_int]
    L.ra_sync.argtypes = [vp]
    L.ra_debug_spectra.argtypes = [vp, vp, ctypes.c_int, vp, vp]
    L.ra_kernel_time.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int),
                                 ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    L.ra_fsc_len.argtypes = [vp]
    L.ra_class_fsc.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, float_ptr]
    L.ra_last_class_fsc.argtypes = [vp, float_ptr]
    L.ra_fit_tanh.argtypes = [float_ptr, float_ptr, ctypes.c_int, float_ptr, float_ptr]
    L.ra_class_averages.argtypes = [vp, vp, vp, ctypes.c_int, vp]
    L.ra_filter_references.argtypes = [vp, vp, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, float_ptr,
                                       ctypes.c_int, float_ptr]
    # reference surface (test_mref_gpu_align.py:95-97 sets the pointer returns to c_ulonglong)
    L.pre_align_init.restype = ctypes.c_ulonglong
    L.pre_align_init.argtypes = [ctypes.c_uint, ctypes.POINTER(AlignConfig), ctypes.c_uint]
    L.pre_align_size_check.restype = ctypes.c_bool
    L.pre_align_size_check.argtypes = [ctypes.c_uint, ctypes.POINTER(AlignConfig), ctypes.c_uint, ctypes.c_float,
                                       ctypes.c_bool]
    L.pre_align_fetch.argtypes = [ctypes.POINTER(float_ptr), ctypes.c_uint, ctypes.c_char_p]
    L.pre_align_fetch.restype = None
    L.pre_align_run.argtypes = [ctypes.c_int, ctypes.c_int]
    L.pre_align_run.restype = None
    L.pre_align_run_m.argtypes = [ctypes.c_int, ctypes.c_int]
    L.pre_align_run_m.restype = ctypes.c_ulonglong
    L.mref_align_run.argtypes = [ctypes.c_int, ctypes.c_int]
    L.mref_align_run.restype = ctypes.c_ulonglong
    L.mref_align_run_m.argtypes = [ctypes.c_int, ctypes.c_int]
    L.mref_align_run_m.restype = float_ptr
    L.get_num_ref.restype = ctypes.POINTER(ctypes.c_int)
    L.reset_shifts.argtypes = [ctypes.c_float, ctypes.c_float]
    L.reset_shifts.restype = None
    L.print_gpu_info.argtypes = [ctypes.c_uint]
    L.print_gpu_info.restype = None
    L.gpu_clear.restype = None
    L.ref_free_alignment_2D_init.restype = ctypes.c_ulonglong
    L.ref_free_alignment_2D_init.argtypes = [ctypes.POINTER(AlignConfig), ctypes.POINTER(float_ptr), ctypes.POINTER(float_ptr),
                                             ctypes.POINTER(ctypes.c_int), ctypes.c_uint]
    L.ref_free_alignment_2D_size_check.restype = ctypes.c_bool
    L.ref_free_alignment_2D_size_check.argtypes = [ctypes.POINTER(AlignConfig), ctypes.c_uint, ctypes.c_float, ctypes.c_bool]
    L.ref_free_alignment_2D.restype = None
    L.ref_free_alignment_2D.argtypes = []
    L.ref_free_alignment_2D_filter_references.restype = None
    L.ref_free_alignment_2D_filter_references.argtypes = [ctypes.c_float, ctypes.c_float]
    L.ra_isac_get_references.argtypes = [float_ptr]
    L.ra_legacy_bytes.restype = ctypes.c_size_t
    L.ra_legacy_bytes.argtypes = [ctypes.c_uint, ctypes.POINTER(AlignConfig)]
    if path is None:
        _lib = L
    return L


def get_c_ptr_array(images):
    """array of float* over C-contiguous float32 images (test_mref_gpu_align.py:138-146)."""
    ptrs = []
    for img in images:
        assert img.flags["C_CONTIGUOUS"] and img.dtype == np.float32
        ptrs.append(img.ctypes.data_as(float_ptr))
    return (float_ptr * len(ptrs))(*ptrs)


def fit_tanh(dres, low=0.1):
    """sp_filter.fit_tanh on an fsc result [freq, fsc, n]; edits dres[1] in place like the original"""
    assert low == 0.1
    n = len(dres[0])
    fr = (ctypes.c_float * n)(*dres[0])
    fs = (ctypes.c_float * n)(*dres[1])
    fl, aa = ctypes.c_float(), ctypes.c_float()
    _check(load_library().ra_fit_tanh(fr, fs, n, ctypes.byref(fl), ctypes.byref(aa)), "ra_fit_tanh")
    dres[1][:] = [float(v) for v in fs]
    return fl.value, aa.value


def _check(rc, what):
    if rc != 0:
        raise EngineError("%s failed (%d): %s" % (what, rc, load_library().ra_last_error().decode()))


def _norm_flag(flag):
    """None or the header's -1: by mode; otherwise on / off (a bare bool(-1) would switch the normalisation ON)"""
    if flag is None or (not isinstance(flag, bool) and isinstance(flag, (int, float)) and flag < 0):
        return -1
    return int(bool(flag))


class Engine:
    """Handle-based engine.  All tensor arguments are torch CUDA tensors on `device`."""

    def __init__(self, nx, last_ring, xrng, yrng, step, nref, mode=RA_MODE_MREF, first_ring=1, ring_skip=1,
                 device=0, chunk=0, interp=RA_INTERP_BILINEAR, normalize_ring=None):
        """interp / normalize_ring: the two hedges of include/ralign.h (ra_options) for the choices of the EMAN2 CPU path that the
        reference tree does not pin -- Util::alrl_ms's interpolation (RA_INTERP_QUADRI: older EMAN2 releases; size-generic kernels)
        and Normalize_ring on / off independent of the mode (None, or the header's -1: by mode)."""
        import torch
        self.torch = torch
        self.lib = load_library()
        self.cfg = RaConfig(int(nx), int(first_ring), int(last_ring), int(ring_skip), float(xrng), float(yrng),
                            float(step), int(nref), int(mode), int(device), int(chunk))
        self.handle = ctypes.c_void_p()
        if interp == RA_INTERP_BILINEAR and _norm_flag(normalize_ring) < 0:
            _check(self.lib.ra_create(ctypes.byref(self.handle), ctypes.byref(self.cfg)), "ra_create")
        else:
            opt = RaOptions(int(interp), _norm_flag(normalize_ring))
            _check(self.lib.ra_create_ex(ctypes.byref(self.handle), ctypes.byref(self.cfg), ctypes.byref(opt)), "ra_create_ex")
        self.nx, self.nref, self.mode, self.device = int(nx), int(nref), int(mode), int(device)
        self.dev = torch.device("cuda", device)

    def close(self):
        if self.handle:
            self.lib.ra_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers
    def _ptr(self, t, dtype=None):
        if t is None:
            return None
        assert t.is_cuda and t.is_contiguous(), "device tensor must be a contiguous CUDA tensor"
        if dtype is not None:
            assert t.dtype == dtype, (t.dtype, dtype)
        return ctypes.c_void_p(t.data_ptr())

    def use_current_stream(self):
        s = self.torch.cuda.current_stream(self.dev)
        _check(self.lib.ra_set_stream(self.handle, ctypes.c_void_p(s.cuda_stream)), "ra_set_stream")

    @property
    def num_shifts(self):
        return self.lib.ra_num_shifts(self.handle)

    @property
    def maxrin(self):
        return self.lib.ra_maxrin(self.handle)

    @property
    def search_path(self):
        """1 = fused particle-resident kernel, 0 = polar + contraction kernel pair, 2 = size-generic kernels"""
        return self.lib.ra_search_path(self.handle)

    @property
    def lcirc(self):
        return self.lib.ra_lcirc(self.handle)

    def new_state(self, n):
        return self.torch.zeros((n, 2), dtype=self.torch.float32, device=self.dev)

    def new_result(self, n):
        return self.torch.zeros((n, 8), dtype=self.torch.int32, device=self.dev)

    @staticmethod
    def result_to_numpy(result):
        return result.cpu().numpy().view(RESULT_DTYPE).reshape(-1)

    # -- API
    def reset_shifts(self, xrng, yrng, step):
        _check(self.lib.ra_reset_shifts(self.handle, xrng, yrng, step), "ra_reset_shifts")

    def set_mask(self, mask):
        """user mask [nx][nx] (CUDA tensor) in place of model_circle(last_ring)"""
        assert mask.shape == (self.nx, self.nx)
        _check(self.lib.ra_set_mask(self.handle, self._ptr(mask, self.torch.float32)), "ra_set_mask")

    def set_normalize_ring(self, flag):
        """Normalize_ring for subsequent searches: True / False, None (or -1) = the mode's default (ra_set_normalize_ring)"""
        _check(self.lib.ra_set_normalize_ring(self.handle, _norm_flag(flag)), "ra_set_normalize_ring")

    @property
    def options(self):
        """(interp, normalize_ring) in force (ra_get_options)"""
        o = RaOptions()
        _check(self.lib.ra_get_options(self.handle, ctypes.byref(o)), "ra_get_options")
        return int(o.interp), int(o.normalize_ring)

    def set_nomirror(self, flag):
        """--nomirror: search the straight orientation only (ormq -> Util.Crosrng_ns)"""
        _check(self.lib.ra_set_nomirror(self.handle, int(bool(flag))), "ra_set_nomirror")

    def set_references(self, refs):
        assert refs.shape == (self.nref, self.nx, self.nx)
        _check(self.lib.ra_set_references(self.handle, self._ptr(refs, self.torch.float32)), "ra_set_references")

    def prepared_references(self):
        out = np.zeros((self.nref, self.lcirc), np.float32)
        _check(self.lib.ra_get_prepared_references(self.handle, out.ctypes.data_as(ctypes.c_void_p)),
               "ra_get_prepared_references")
        return out

    def align(self, particles, state, result, cs=None):
        n = particles.shape[0]
        assert particles.shape[1:] == (self.nx, self.nx) and state.shape == (n, 2) and result.shape == (n, 8)
        csp = None
        if cs is not None:
            csp = (ctypes.c_float * 2)(float(cs[0]), float(cs[1]))
        _check(self.lib.ra_align(self.handle, self._ptr(particles, self.torch.float32), n,
                                 self._ptr(state, self.torch.float32), self._ptr(result, self.torch.int32), csp),
               "ra_align")

    @property
    def search_tiled(self):
        return bool(self.lib.ra_search_tiled(self.handle))

    @property
    def search_skips_offsets(self):
        """True when the search evaluates the in-window offsets of a particle only (ra_search_skips_offsets)"""
        return bool(self.lib.ra_search_skips_offsets(self.handle))

    @property
    def search_offsets_per_pass(self):
        """search_path == 3 (rings of 512 samples): 2 = search_duo_kernel, 1 = search_solo_kernel; 0 otherwise"""
        return self.lib.ra_search_offsets_per_pass(self.handle)

    def set_refine(self, threshold):
        """sub-bin angle refinement with the CPU path's arithmetic (ra_set_refine): threshold on |c3| / max |b| of prb1d,
        < 0 = every particle, 0 = off; call before set_references"""
        _check(self.lib.ra_set_refine(self.handle, float(threshold)), "ra_set_refine")

    def state_from_params(self, result, state, cs=None):
        """the reference's state round trip (ra_state_from_params): state <- inverse_transform2 of the float32 parameters
        in `result` (with the centre correction cs folded in first in the reference-free mode)"""
        n = result.shape[0]
        assert state.shape == (n, 2) and result.shape == (n, 8)
        csp = None
        if cs is not None:
            csp = (ctypes.c_float * 2)(float(cs[0]), float(cs[1]))
        _check(self.lib.ra_state_from_params(self.handle, self._ptr(result, self.torch.int32), n, csp,
                                             self._ptr(state, self.torch.float32)), "ra_state_from_params")

    def last_refine_count(self):
        """particles the last search launch re-evaluated in the CPU path's arithmetic (diagnostics; synchronises)"""
        return self.lib.ra_last_refine_count(self.handle)

    def state_from_params_dev(self, result, state, cs_dev):
        """state_from_params with the centre correction in a CUDA tensor [2] (no host value in the path)"""
        n = result.shape[0]
        assert state.shape == (n, 2) and result.shape == (n, 8) and cs_dev.numel() == 2
        _check(self.lib.ra_state_from_params_dev(self.handle, self._ptr(result, self.torch.int32), n,
                                                 self._ptr(cs_dev, self.torch.float32), self._ptr(state, self.torch.float32)),
               "ra_state_from_params_dev")

    def set_class_references(self, refs):
        """class-resident mode: one reference per class, [ncls][nx][nx] (ra_set_class_references)"""
        assert refs.shape[1:] == (self.nx, self.nx)
        _check(self.lib.ra_set_class_references(self.handle, self._ptr(refs, self.torch.float32), int(refs.shape[0])),
               "ra_set_class_references")

    def align_classes(self, particles, state, result, cls):
        """every particle against the reference of its class cls[i] (int32, device), one launch (ra_align_classes)"""
        n = particles.shape[0]
        assert particles.shape[1:] == (self.nx, self.nx) and state.shape == (n, 2) and result.shape == (n, 8) and cls.shape == (n,)
        _check(self.lib.ra_align_classes(self.handle, self._ptr(particles, self.torch.float32), n,
                                         self._ptr(state, self.torch.float32), self._ptr(result, self.torch.int32),
                                         self._ptr(cls, self.torch.int32)), "ra_align_classes")

    def transform_accumulate(self, particles, result, index0=0, aligned=None, sums=None, counts=None):
        n = particles.shape[0]
        _check(self.lib.ra_transform_accumulate(self.handle, self._ptr(particles, self.torch.float32), n, int(index0),
                                                self._ptr(result, self.torch.int32),
                                                self._ptr(aligned, self.torch.float32),
                                                self._ptr(sums, self.torch.float32),
                                                self._ptr(counts, self.torch.int32)), "ra_transform_accumulate")

    def update_references(self, sums, counts, refs, min_count=4):
        _check(self.lib.ra_update_references(self.handle, self._ptr(sums, self.torch.float32),
                                             self._ptr(counts, self.torch.int32), int(min_count),
                                             self._ptr(refs, self.torch.float32)), "ra_update_references")

    def normalize_particles(self, particles):
        _check(self.lib.ra_normalize_particles(self.handle, self._ptr(particles, self.torch.float32),
                                               particles.shape[0]), "ra_normalize_particles")

    def debug_spectra(self, particles, state):
        """ring spectra [n][num_shifts][lcirc] (EMAN2 packing) of the polar / ring-FFT stage alone"""
        n = particles.shape[0]
        out = np.zeros((n, self.num_shifts, self.lcirc), np.float32)
        _check(self.lib.ra_debug_spectra(self.handle, self._ptr(particles, self.torch.float32), n,
                                         self._ptr(state, self.torch.float32), out.ctypes.data_as(ctypes.c_void_p)),
               "ra_debug_spectra")
        return out

    # -- reference update on the device
    def class_fsc(self, sums, counts, min_count=4, masked=False):
        """[freq, fsc, npoints] lists like sp_statistics.fsc / fsc_mask, averaged over the live classes"""
        n = self.lib.ra_fsc_len(self.handle)
        out = np.zeros((3, n), np.float32)
        _check(self.lib.ra_class_fsc(self.handle, self._ptr(sums, self.torch.float32), self._ptr(counts, self.torch.int32),
                                     int(min_count), int(bool(masked)), out.ctypes.data_as(float_ptr)), "ra_class_fsc")
        return [list(map(float, out[0])), list(map(float, out[1])), list(map(float, out[2]))]

    def last_class_fsc(self):
        """per-class curves of the last class_fsc call: (fsc [nref][len], points per shell [nref][len]) (ra_last_class_fsc)"""
        n = self.lib.ra_fsc_len(self.handle)
        out = np.zeros((self.nref, 2, n), np.float32)
        _check(self.lib.ra_last_class_fsc(self.handle, out.ctypes.data_as(float_ptr)), "ra_last_class_fsc")
        return out[:, 0], out[:, 1]

    def class_averages(self, sums, counts, refs, min_count=4):
        _check(self.lib.ra_class_averages(self.handle, self._ptr(sums, self.torch.float32),
                                          self._ptr(counts, self.torch.int32), int(min_count),
                                          self._ptr(refs, self.torch.float32)), "ra_class_averages")

    def class_fsc_fit(self, sums, counts, fit, curve, min_count=4, masked=False, fl_lo=0.12, fl_hi=0.4, aa_hi=0.2):
        """class_fsc + fit_tanh + the clamps of ref_ali2d on the device: fit [5] and curve [3][fsc_len] are CUDA float tensors the
        kernels fill (ra_class_fsc_fit); nothing is read back here"""
        n = self.lib.ra_fsc_len(self.handle)
        assert fit.numel() >= 5 and curve.numel() >= 3 * n
        _check(self.lib.ra_class_fsc_fit(self.handle, self._ptr(sums, self.torch.float32), self._ptr(counts, self.torch.int32),
                                         int(min_count), int(bool(masked)), float(fl_lo), float(fl_hi), float(aa_hi),
                                         self._ptr(fit, self.torch.float32), self._ptr(curve, self.torch.float32)), "ra_class_fsc_fit")

    def filter_references_dev(self, imgs, flaa=None, center=0, cs_in=None, normalize=True, cs_out=None):
        """filter_references with (fl, aa), the centres of center = -1 and the applied centres in CUDA tensors; asynchronous"""
        m = imgs.shape[0]
        _check(self.lib.ra_filter_references_dev(self.handle, self._ptr(imgs, self.torch.float32), m,
                                                 self._ptr(flaa, self.torch.float32), int(center), self._ptr(cs_in, self.torch.float32),
                                                 int(bool(normalize)), self._ptr(cs_out, self.torch.float32)), "ra_filter_references_dev")

    @property
    def fsc_len(self):
        return self.lib.ra_fsc_len(self.handle)

    def filter_references(self, imgs, fl, aa, center=0, cs_in=None, normalize=True):
        """in place on imgs [m][nx][nx]; returns the applied centres [m][2]"""
        m = imgs.shape[0]
        cin = None
        if cs_in is not None:
            cin = np.ascontiguousarray(cs_in, np.float32).reshape(m, 2)
        cout = np.zeros((m, 2), np.float32)
        _check(self.lib.ra_filter_references(self.handle, self._ptr(imgs, self.torch.float32), m, float(fl), float(aa),
                                             int(center), cin.ctypes.data_as(float_ptr) if cin is not None else None,
                                             int(bool(normalize)), cout.ctypes.data_as(float_ptr)), "ra_filter_references")
        return cout

    def sync(self):
        _check(self.lib.ra_sync(self.handle), "ra_sync")

    def kernel_time(self, enable=True):
        """returns (ms_ccf, launches_ccf, ms_polar, launches_polar) since the last call and
        (re)arms HIP-event timing of the two hot kernels on the engine's stream."""
        a = ctypes.c_double(); b = ctypes.c_double(); na = ctypes.c_int(); nb = ctypes.c_int()
        _check(self.lib.ra_kernel_time(self.handle, int(enable), ctypes.byref(a), ctypes.byref(na), ctypes.byref(b),
                                       ctypes.byref(nb)), "ra_kernel_time")
        return a.value, na.value, b.value, nb.value
