"""CPU tests of the host logic: geometry helpers, the C-ABI library (loads, exports every
symbol include/ralign.h declares, refuses to compute without a GPU), the build entry."""
import ctypes
import os
import re

import numpy as np
import pytest

from cryo_ralib_amd import api, geometry, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mpi_start_end_partition():
    for total, world in [(50000, 8), (1000, 3), (7, 4), (5, 8)]:
        edges = [geometry.mpi_start_end(total, world, r) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == total
        for a, b in zip(edges[:-1], edges[1:]):
            assert a[1] == b[0]
    assert geometry.mpi_start_end(10, 4, 1) == (2, 5)      # round(2.5)=2 (banker's), round(5.0)=5


def test_shift_list_order():
    s = geometry.shift_list(3, 3, 1)
    assert s.shape == (49, 2)
    assert tuple(s[0]) == (-3, -3) and tuple(s[1]) == (-2, -3) and tuple(s[7]) == (-3, -2) and tuple(s[-1]) == (3, 3)
    assert geometry.shift_list(1, 2, 0.5).shape == (5 * 9, 2)


def test_alignparam_to_eman2_matches_reference_tail():
    # test_mref_gpu_align.py:578-588
    ang, sx, sy, m = geometry.alignparam_to_eman2([90.0], [2.0], [1.0], [1])
    assert ang[0] == 90.0 and m[0] == 1
    # co = 0, so = -1: sx = (-2)*0 - (-1)*(-1) = -1 ; sy = (-2)*(-1) + (-1)*0 = 2
    assert sx[0] == pytest.approx(-1.0, abs=1e-12) and sy[0] == pytest.approx(2.0, abs=1e-12)


def test_synthetic_generator_is_seeded():
    r1 = synth.make_references(3, 32, 12)
    r2 = synth.make_references(3, 32, 12)
    np.testing.assert_array_equal(r1, r2)
    p1, t1 = synth.make_particles(r1, 5, 2, 2, 0.5, shard=1)
    p2, t2 = synth.make_particles(r1, 5, 2, 2, 0.5, shard=1)
    np.testing.assert_array_equal(p1, p2)
    p3, _ = synth.make_particles(r1, 5, 2, 2, 0.5, shard=2)
    assert np.abs(p1 - p3).max() > 0
    m = geometry.model_circle(12, 32, 32)
    sel = m > 0.5
    assert abs(r1[0][sel].mean()) < 1e-5 and r1[0][sel].std(ddof=1) == pytest.approx(1, abs=1e-4)


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "ralign.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b([A-Za-z_0-9]+)\s*\(", txt)
    return sorted(set(n for n in names if n.startswith("ra_") or n in (
        "print_gpu_info", "gpu_clear", "pre_align_init", "pre_align_size_check", "pre_align_fetch", "pre_align_run",
        "pre_align_run_m", "mref_align_run", "mref_align_run_m", "get_num_ref", "reset_shifts",
        "ref_free_alignment_2D_init", "ref_free_alignment_2D_size_check", "ref_free_alignment_2D",
        "ref_free_alignment_2D_filter_references")))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    lib = ctypes.CDLL(api.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 27
    assert sorted(api.EXPORTED_SYMBOLS) == syms
    for s in syms:
        assert hasattr(lib, s), s


def test_shipped_library_reads_only_the_documented_switches():
    """experiment switches (LDS strides, job orders, block shapes) exist only in builds with
    -DRALIGN_PROFILE_SWITCHES; the shipped binary holds exactly the RALIGN_* names README.md documents as user-facing"""
    with open(api.LIB_PATH, "rb") as f:
        blob = f.read()
    names = sorted(set(m.decode() for m in re.findall(rb"RALIGN_[A-Z0-9_]+", blob)))
    documented = ["RALIGN_ATOMIC_SUMS", "RALIGN_CROP", "RALIGN_DUO", "RALIGN_FUSED", "RALIGN_GCCF_SPLIT", "RALIGN_GCCF_TM",
                  "RALIGN_GENERIC", "RALIGN_GRID", "RALIGN_INFO", "RALIGN_LIVE_OFFSETS", "RALIGN_PACK", "RALIGN_PAIR", "RALIGN_REFINE", "RALIGN_SOLO",
                  "RALIGN_SOLO_JOBS", "RALIGN_TCROP", "RALIGN_TIGHT_RINGS", "RALIGN_TILED", "RALIGN_XSUM", "RALIGN_XTILE", "RALIGN_ZONES"]
    assert names == documented, sorted(set(names) ^ set(documented))
    readme = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "README.md")).read()
    for n in documented:
        assert "`%s" % n in readme, n


def test_struct_layouts_match_reference_abi():
    # cuda/gpu_aln_common.h:62-83: 8 x 4 B and 24 B
    assert ctypes.sizeof(api.AlignConfig) == 32
    assert ctypes.sizeof(api.AlignParam) == 24
    assert api.AlignParam.mirror.offset == 20
    assert ctypes.sizeof(api.RaConfig) == 44
    assert api.RESULT_DTYPE.itemsize == 32


def test_engine_fails_loudly_without_gpu_or_library(tmp_path):
    import torch
    with pytest.raises(api.EngineError):
        api.load_library(str(tmp_path / "missing.so"))
    if not torch.cuda.is_available():
        with pytest.raises(api.EngineError) as ei:
            api.Engine(90, 36, 3, 3, 1.0, 10)
        assert "ra_create failed" in str(ei.value)


def test_ra_create_rejects_bad_geometry():
    lib = api.load_library()
    h = ctypes.c_void_p()
    # "Shift or radius is too large - particle crosses image boundary" (test_mref_gpu_align.py:314)
    cfg = api.RaConfig(90, 1, 43, 1, 3.0, 3.0, 1.0, 10, 0, 0, 0)
    assert lib.ra_create(ctypes.byref(h), ctypes.byref(cfg)) == -1
    assert b"crosses image boundary" in lib.ra_last_error()
    cfg = api.RaConfig(90, 1, 36, 1, 3.0, 3.0, 1.0, 0, 0, 0, 0)
    assert lib.ra_create(ctypes.byref(h), ctypes.byref(cfg)) == -1
    assert lib.ra_create(None, None) == -1


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "cryo_ralib_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("# oracle-free", ""), f


def test_build_tracks_every_engine_header():
    """a header that is not in build.HEADERS can change without a rebuild: the library then runs stale code"""
    import re
    from cryo_ralib_amd import build
    csrc = os.path.join(os.path.dirname(build.__file__), "csrc")
    tracked = {os.path.basename(h) for h in build.HEADERS}
    included = set()
    for name in os.listdir(csrc):
        with open(os.path.join(csrc, name)) as f:
            included.update(re.findall(r'#include "([^"/]+\.h)"', f.read()))
    assert included and included <= tracked, sorted(included - tracked)
    assert all(os.path.exists(h) for h in build.HEADERS)


def test_stack_io_round_trip(tmp_path):
    from cryo_ralib_amd import stackio
    rng = np.random.default_rng(0)
    a = rng.normal(size=(5, 12, 12)).astype(np.float32)
    for ext in ("npy", "mrcs", "hdf"):
        p = str(tmp_path / ("s." + ext))
        stackio.write_stack(p, a)
        np.testing.assert_array_equal(stackio.read_stack(p), a)
    with pytest.raises((RuntimeError, ValueError, OSError)):
        stackio.read_stack(str(tmp_path / "missing.hdf"))
    stackio.write_text_rows(str(tmp_path / "p.txt"), [(0, 1.968414, 1.102456, 2.963881, 0, 0)])
    row = open(tmp_path / "p.txt").read().split()
    assert [float(x) for x in row] == [0, 1.968414, 1.102456, 2.963881, 0, 0]      # notebook/03 cell 6 column order


def test_cli_flags_match_reference():
    from cryo_ralib_amd import cli
    import argparse
    # test_mref_gpu_align.py:1142-1159 and test_reffree_gpu_align.py:918-935 flag names
    src = open(os.path.join(ROOT, "cryo_ralib_amd", "cli.py")).read()
    for flag in ("--ir", "--ou", "--rs", "--xr", "--yr", "--ts", "--center", "--maxit", "--CTF", "--snr", "--function",
                 "--rand_seed", "--gpu_devices", "--gpu_info", "--MPI", "--EQ", "--nomirror", "--dst", "--Fourvar",
                 "--mode", "--random_method"):
        assert '"%s"' % flag in src, flag
    assert cli._first("4 2 1 1") == 4.0 and cli._first(3) == 3.0


def test_fit_tanh_matches_the_python_restatement():
    """ra_fit_tanh (C++, in the product library; pure host arithmetic) against oracle/refine_oracle.fit_tanh
    on FSC curves of different shapes, including the early-exit branches"""
    import math
    from cryo_ralib_amd import api
    from oracle import refine_oracle as ro
    n = 46
    freq = [i / (2.0 * (n - 1)) for i in range(n)]

    def halfset(full):      # the fit converts half-set fsc r to 2r/(1+r): invert so that the fitted curve is `full`
        return [f / (2.0 - f) for f in full]

    curves = []
    for fl, aa in ((0.15, 0.1), (0.25, 0.3), (0.08, 0.2), (0.35, 0.05)):
        full = [0.5 * (math.tanh(math.pi * (f + fl) / 2.0 / aa / fl) - math.tanh(math.pi * (f - fl) / 2.0 / aa / fl)) for f in freq]
        curves.append(halfset(full))
    curves.append([1.0] * n)                              # never falls: (0.49, 0.1)
    curves.append([0.9] * (n - 1) + [0.2])                # falls only at the last point: (0.5, 0.2)
    rng = __import__("numpy").random.default_rng(3)
    noisy = [max(-0.2, min(1.0, c + 0.05 * rng.standard_normal())) for c in curves[0]]
    noisy[0] = -0.3                                       # the sign flip of the first point
    curves.append(noisy)
    for c in curves:
        a = [list(freq), [float(__import__("numpy").float32(v)) for v in c], [2.0] * n]
        b = [list(a[0]), list(a[1]), list(a[2])]
        want = ro.fit_tanh(a)
        got = api.fit_tanh(b)
        assert abs(got[0] - want[0]) < 2e-5 and abs(got[1] - want[1]) < 2e-5, (got, want)
        assert max(abs(x - y) for x, y in zip(a[1], b[1])) < 1e-7      # same in-place edit of the curve


def test_refine_oracle_known_properties():
    """the numpy restatement of fsc / filt_tanl / phase_cog / fshift behaves as the definitions say"""
    import numpy as np
    from oracle import refine_oracle as ro
    rng = np.random.default_rng(5)
    nx = 32
    a = rng.standard_normal((nx, nx)).astype(np.float32)
    f = ro.fsc(a, a)
    assert len(f[0]) == nx // 2 + 1 and abs(f[0][-1] - 0.5) < 1e-12
    assert all(abs(v - 1.0) < 1e-6 for v in f[1])                      # identical halves correlate perfectly
    b = rng.standard_normal((nx, nx)).astype(np.float32)
    assert max(abs(v) for v in ro.fsc(a, b)[1][3:]) < 0.6              # independent noise does not
    # tangent filter: DC passes, Nyquist is suppressed; pinned formula of cuda/gpu_aln_noref.cu:799-814
    h = ro.tanl_filter_values(nx, 0.12, 0.2)
    assert abs(h[0, 0] - 1.0) < 1e-3 and h[0, nx // 2] < 1e-3
    # a blob displaced from the centre by (+3, -2): phase_cog finds it, fshift(-cs) puts it back
    yy, xx = np.mgrid[0:nx, 0:nx]
    blob = np.exp(-((xx - nx // 2 - 3) ** 2 + (yy - nx // 2 + 2) ** 2) / 18.0).astype(np.float32)
    cs = ro.phase_cog(blob)
    assert abs(cs[0] - 3) < 0.05 and abs(cs[1] + 2) < 0.05
    back = ro.fshift(blob, -cs[0], -cs[1])
    c2 = ro.phase_cog(back)
    assert abs(c2[0]) < 1e-3 and abs(c2[1]) < 1e-3
    # amoeba maximises
    best, val, _ = ro.amoeba([0.0, 0.0], [0.5, 0.5], lambda p, d: -((p[0] - 1) ** 2 + (p[1] + 2) ** 2))
    assert abs(best[0] - 1) < 1e-2 and abs(best[1] + 2) < 1e-2


def test_mdf_hdf5_reader_on_a_file_written_by_libhdf5(golden_dir):
    """tests/golden/eman2_mdf_sample.hdf was produced with the HDF5 C library (make_hdf_fixture.c): 37 images in the
    EMAN2 MDF layout (more than one symbol-table node per group), value(i, y, x) = 1000 i + 16 y + x + 0.25"""
    import os
    import numpy as np
    from cryo_ralib_amd import mdfio, stackio
    path = os.path.join(golden_dir, "eman2_mdf_sample.hdf")
    a, attrs = mdfio.read_mdf_stack(path, with_attrs=True)
    assert a.shape == (37, 10, 12) and a.dtype == np.float32
    i, y, x = np.mgrid[0:37, 0:10, 0:12]
    assert np.array_equal(a, (1000 * i + 16 * y + x + 0.25).astype(np.float32))
    assert attrs[5]["EMAN.nx"] == 12 and attrs[5]["EMAN.ny"] == 10 and attrs[5]["EMAN.ptcl_repr"] == 2
    assert abs(float(attrs[36]["EMAN.apix_x"]) - 1.25) < 1e-7
    assert np.array_equal(stackio.read_stack(path), a)


def test_mdf_hdf5_writer_round_trip_and_libhdf5_check(tmp_path):
    """the writer's files read back bit for bit; where the HDF5 command line tools exist (the build container's
    /opt/conda) the real library must list every object and print the same pixels"""
    import shutil
    import subprocess
    import numpy as np
    from cryo_ralib_amd import mdfio, stackio
    rng = np.random.default_rng(11)
    for n in (1, 9, 70, 1300):                      # 1 .. several B-tree levels at the library's default node sizes
        arr = rng.standard_normal((n, 6, 5)).astype(np.float32)
        path = str(tmp_path / ("s%d.hdf" % n))
        stackio.write_stack(path, arr)
        assert np.array_equal(stackio.read_stack(path), arr)
        h5ls = shutil.which("h5ls") or "/opt/conda/bin/h5ls"
        h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
        if not (shutil.os.path.exists(h5ls) and shutil.os.path.exists(h5dump)):
            continue
        out = subprocess.run([h5ls, path + "/MDF/images"], capture_output=True, text=True)
        assert out.returncode == 0 and len(out.stdout.strip().splitlines()) == n, out.stderr
        out = subprocess.run([h5dump, "-d", "/MDF/images/%d/image" % (n - 1), "-y", "-w", "0", path], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        body = out.stdout.split("DATA {")[1].split("}")[0].replace("\\n", " ")
        vals = np.array([float(v) for v in body.replace(",", " ").split()], np.float32)
        np.testing.assert_allclose(vals, arr[n - 1].ravel(), rtol=1e-5, atol=1e-6)
        out = subprocess.run([h5dump, "-a", "/MDF/images/imageid_max", path], capture_output=True, text=True)
        assert out.returncode == 0 and ("(0): %d" % (n - 1)) in out.stdout
    with __import__("pytest").raises(mdfio.HDF5FormatError):
        p = tmp_path / "bad.hdf"
        p.write_bytes(b"not hdf5" * 100)
        mdfio.read_mdf_stack(str(p))


def test_reference_binding_statements_run_unchanged(golden_dir):
    """every `cu_module.<symbol>.restype = ...` statement the reference's drivers execute at import
    (test_mref_gpu_align.py:95-97, test_reffree_gpu_align.py:97-99, test_reffree.py:56-57,
    test_mref_cheng_yu_bdb_cuda.py:60-63) and every library function they call resolves in libralign_hip.so.
    The statement list is data collected from the reference tree by tests/golden/make_binding_fixture.py."""
    import json
    cu_module = ctypes.CDLL(api.LIB_PATH)     # what `ctypes.CDLL(CUDA_PATH + "gpu_aln_pack.so")` becomes
    fx = json.load(open(os.path.join(golden_dir, "binding_symbols.json")))
    assert set(fx) == {"test_mref_gpu_align.py", "test_reffree_gpu_align.py", "test_reffree.py", "test_mref_cheng_yu_bdb_cuda.py"}
    env = {"ctypes": ctypes, "cu_module": cu_module}
    for driver, d in fx.items():
        assert d["restype_statements"], driver
        for st in d["restype_statements"]:
            exec("cu_module.%s.restype = %s" % (st["symbol"], st["restype"]), env)      # AttributeError if not exported
        for name in d["called"]:
            assert hasattr(cu_module, name), (driver, name)
    assert cu_module.ref_free_alignment_2D_init.restype is ctypes.c_ulonglong


def test_tangent_filter_profile_matches_the_reference_tree(golden_dir):
    """H(d) of the oracle's filt_tanl against the output of the reference tree's kernel text
    (cuda/gpu_aln_noref.cu:786-816, run on the host by tests/golden/make_reftree_pins.py)"""
    from oracle import refine_oracle as ro
    t = np.load(os.path.join(golden_dir, "tanl_ref.npz"))
    assert int(t["ncase"]) == 6
    for k in range(int(t["ncase"])):
        nx, fl, aa = t["par%d" % k]
        got = ro.tanl_profile(t["d%d" % k].astype(np.float64), fl, aa)
        assert np.abs(got - t["H%d" % k]).max() < 5e-7       # float tanhf vs double tanh
    # and the filter the oracle applies is that profile on EMAN2's frequency grid (kx/nx, ky/ny)
    H = ro.tanl_filter_values(90, 0.2, 0.1)
    assert H.shape == (90, 46) and abs(H[0, 0] - 1.0) < 1e-6 and H[45, 45] < 1e-6
    assert abs(H[0, 18] - 0.5) < 1e-6            # |k| = fl = 18/90: the half-height point


def test_filter_clamps_match_the_reference_run_log(golden_dir):
    """notebook/00 cell 3 logs "cut-off 0.120 / fall-off 0.200": ref_ali2d clamps fl to [0.12, 0.4] and aa to <= 0.2"""
    import json
    from oracle import refine_oracle as ro
    rows = json.load(open(os.path.join(golden_dir, "filter_log.json")))["cutoff_falloff"]
    assert [0.12, 0.2] in rows
    assert min(r[0] for r in rows) == 0.12 and max(r[1] for r in rows) == 0.2
    # an FSC that drops at once drives fit_tanh below the lower clamp and above the fall-off clamp
    n = 46
    freq = [i / 90.0 for i in range(n)]
    fsc = [1.0, 0.9, 0.3, 0.05] + [0.0] * (n - 4)
    img = np.zeros((90, 90), np.float32); img[45, 45] = 1.0
    _, _, fl, aa = ro.ref_ali2d(None, 0, img, [freq, fsc, [1.0] * n])
    assert fl == 0.12 and aa <= 0.2
    f2, a2 = api.fit_tanh([list(freq), list(fsc), [1.0] * n])
    assert max(min(0.4, f2), 0.12) == 0.12


def test_bench_gpus_flag_never_runs_a_single_rank_silently():
    """`bench.py --gpus 2` without a launcher starts 2 ranks itself (before any GPU call); here, without a GPU, both
    ranks must fail loudly -- the one thing it may not do is print a 1-GPU line.  A launcher with another world size is
    rejected."""
    import subprocess
    import sys
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--particles", "8", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0
    assert '"n_gpus": 1' not in r.stdout
    env["WORLD_SIZE"] = "4"; env["RANK"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "launcher started 4 ranks" in (r.stderr + r.stdout)


def test_header_write_back_and_slice_reads(tmp_path):
    """xform.align2d / assign / ID land in the MDF headers (test_mref_cheng_yu_bdb_cuda.py:114-203) and a rank can
    read its MPI_start_end slice without the rest of the file"""
    from cryo_ralib_amd import mdfio, stackio
    rng = np.random.default_rng(5)
    imgs = rng.normal(size=(23, 16, 16)).astype(np.float32)
    path = str(tmp_path / "stack.hdf")
    mdfio.write_mdf_stack(path, imgs, [{"ptcl_source_coord": np.int32(i * 7)} for i in range(23)])
    assert stackio.stack_size(path) == 23
    for lo, hi in [(0, 23), (5, 11), (22, 23), (7, 8)]:
        np.testing.assert_array_equal(stackio.read_stack(path, lo, hi), imgs[lo:hi])
    params = [(float(rng.uniform(0, 360)), float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)), int(rng.integers(0, 2))) for _ in range(23)]
    assign = list(rng.integers(0, 5, 23))
    mdfio.write_alignment_headers(path, path, params, assign=assign, ids=list(range(100, 123)))
    arr, attrs = mdfio.read_mdf_stack(path, with_attrs=True)
    np.testing.assert_array_equal(arr, imgs)
    for i in range(23):
        a = attrs[i]
        assert int(a["EMAN.assign"]) == assign[i] and int(a["EMAN.ID"]) == 100 + i and int(a["EMAN.ptcl_source_coord"]) == 7 * i
        m = a["EMAN.xform.align2d"]
        assert m.shape == (12,)
        al, sx, sy, mir = mdfio.params_from_matrix(m)
        assert mir == params[i][3] and abs(sx - params[i][1]) < 1e-5 and abs(sy - params[i][2]) < 1e-5
        assert abs(((al - params[i][0]) + 180) % 360 - 180) < 1e-4
    # the matrix follows the transform the oracle's parameter algebra uses: R(alpha) = [[c, s], [-s, c]]
    m = mdfio.transform2d_matrix(30.0, 1.5, -2.0, 0).reshape(3, 4)
    np.testing.assert_allclose(m[:2, :2], [[np.cos(np.pi / 6), 0.5], [-0.5, np.cos(np.pi / 6)]], atol=1e-6)
    # other formats slice as well
    mrc = str(tmp_path / "s.mrcs"); npy = str(tmp_path / "s.npy")
    stackio.write_stack(mrc, imgs); stackio.write_stack(npy, imgs)
    for pth in (mrc, npy):
        assert stackio.stack_size(pth) == 23
        np.testing.assert_array_equal(stackio.read_stack(pth, 3, 9), imgs[3:9])


def test_header_write_back_never_loses_header_items(tmp_path, golden_dir):
    """a stack written by libhdf5 (tests/golden/make_hdf_strings_fixture.c) with a fixed-length string, a double and a
    variable-length string (EMAN.ctf) in every image header: the write-back carries strings and doubles over, names
    what it cannot encode, and refuses to replace the user's stack in place rather than dropping header items"""
    import shutil
    from cryo_ralib_amd import mdfio
    src = str(tmp_path / "in.hdf")
    shutil.copy(os.path.join(golden_dir, "eman2_mdf_strings.hdf"), src)
    before = open(src, "rb").read()
    rep = {}
    arr, attrs = mdfio.read_mdf_stack(src, with_attrs=True, report=rep)
    assert arr.shape == (5, 6, 8) and arr[3, 2, 5] == 3 * 100 + 2 * 8 + 5 + 0.5
    assert rep == {"undecoded_attributes": {"EMAN.ctf"}}
    assert bytes(attrs[2]["EMAN.source_path"]) == b"mics/micrograph_02.hdf" and attrs[2]["EMAN.apix_y"].dtype == np.float64
    params = [(10.0 * i, 1.0, -2.0, i % 2) for i in range(5)]
    with pytest.raises(mdfio.HDF5FormatError, match="EMAN.ctf"):
        mdfio.write_alignment_headers(src, src, params)                  # in place: refused, nothing touched
    assert open(src, "rb").read() == before and not os.path.exists(src + ".tmp")
    dst = str(tmp_path / "out.hdf")
    assert mdfio.write_alignment_headers(src, dst, params, assign=[1, 2, 3, 4, 0]) == ["EMAN.ctf"]
    arr2, at2 = mdfio.read_mdf_stack(dst, with_attrs=True)
    np.testing.assert_array_equal(arr2, arr)
    for i in range(5):
        assert bytes(at2[i]["EMAN.source_path"]) == b"mics/micrograph_%02d.hdf" % i
        assert at2[i]["EMAN.apix_y"] == np.float64(1.2500000001) and at2[i]["EMAN.apix_x"] == np.float32(1.25)
        assert int(at2[i]["EMAN.source_n"]) == 3 * i and int(at2[i]["EMAN.assign"]) == [1, 2, 3, 4, 0][i]
        assert mdfio.params_from_matrix(at2[i]["EMAN.xform.align2d"])[3] == i % 2
    # a stack without such items is still replaced in place (the reference's behaviour, --header_writeback)
    plain = str(tmp_path / "plain.hdf")
    mdfio.write_mdf_stack(plain, arr, [{"source_path": "a/b.hdf", "ptcl_source_coord": np.array([3, 4], np.int32)} for _ in range(5)])
    assert mdfio.write_alignment_headers(plain, plain, params) == []
    _, at3 = mdfio.read_mdf_stack(plain, with_attrs=True)
    assert bytes(at3[4]["EMAN.source_path"]) == b"a/b.hdf" and list(at3[4]["EMAN.ptcl_source_coord"]) == [3, 4]
    # ... and a string attribute keeps its stored size over repeated write-backs (value + terminating NUL)
    size1, size_after_first = at3[4]["EMAN.source_path"].dtype.itemsize, os.path.getsize(plain)
    for _ in range(3):
        assert mdfio.write_alignment_headers(plain, plain, params) == []
    _, at4 = mdfio.read_mdf_stack(plain, with_attrs=True)
    assert at4[4]["EMAN.source_path"].dtype.itemsize == size1 <= len(b"a/b.hdf") + 1 and bytes(at4[4]["EMAN.source_path"]) == b"a/b.hdf"
    assert os.path.getsize(plain) == size_after_first


@pytest.mark.parametrize("argv", [["--center=3"], ["--CTF"], ["--random_method=SHC"], ["--Fourvar"], ["--mode=H"], ["--dst=90"],
                                  ["--randomize"], ["--orient"]])
def test_command_line_rejects_what_the_engine_does_not_implement(argv, tmp_path):
    """options of the reference's command lines that would change the result and are not implemented end the run with an
    error before anything is read or computed (test_reffree_gpu_align.py:918-935, test_mref_gpu_align.py:1146-1152)"""
    from cryo_ralib_amd import cli
    with pytest.raises(SystemExit) as e:
        cli.main_reffree([str(tmp_path / "nostack.hdf"), str(tmp_path / "out")] + argv)
    assert "not implemented" in str(e.value)
    if argv[0] in ("--center=3", "--CTF"):
        with pytest.raises(SystemExit) as e:
            cli.main_mref([str(tmp_path / "nostack.hdf"), str(tmp_path / "norefs.hdf"), str(tmp_path / "out")] + argv)
        assert "not implemented" in str(e.value)
