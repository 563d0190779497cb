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
    names = re.findall(r"\b([a-z_0-9]+)\s*\(", txt)
    return sorted(set(n for n in names if n.startswith("ra_") or n in (
        "print_gpu_info", "gpu_clear", "pre_align_init", "pre_align_size_check", "pre_align_fetch", "pre_align_run",
        "pre_align_run_m", "mref_align_run", "mref_align_run_m", "get_num_ref", "reset_shifts")))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    lib = ctypes.CDLL(api.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 27
    assert sorted(api.EXPORTED_SYMBOLS) == syms
    for s in syms:
        assert hasattr(lib, s), s


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


def test_stack_io_round_trip(tmp_path):
    from cryo_ralib_amd import stackio
    rng = np.random.default_rng(0)
    a = rng.normal(size=(5, 12, 12)).astype(np.float32)
    for ext in ("npy", "mrcs"):
        p = str(tmp_path / ("s." + ext))
        stackio.write_stack(p, a)
        np.testing.assert_array_equal(stackio.read_stack(p), a)
    with pytest.raises((RuntimeError, ValueError)):
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
