"""two ranks of the multi-reference driver under RCCL (skipped unless the box has two GPUs): the class sums, counts and
new references are bitwise identical on both ranks and agree with the one-rank run within the order-of-summation bound"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir, nx, ou, nref, xr, n, backend=None):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from cryo_ralib_amd import dist as rdist, synth
    from cryo_ralib_amd.mref import MrefAligner
    r, local, w = rdist.init_from_env(backend)          # default: nccl == RCCL
    local = local % torch.cuda.device_count()
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 0.5)
    lo, hi = rdist.shard_range(n, w, r)
    al = MrefAligner(parts[lo:hi], refs, ou, xr, xr, 1.0, device=local, index0=lo, total_nima=n, preprocess=True)
    al.search()
    al.buf.all_reduce()
    sums, counts = al.buf.sums.cpu().numpy().copy(), al.buf.counts_i.cpu().numpy().copy()
    al.engine.update_references(al.buf.sums, al.buf.counts_i, al.refs, 4)
    al.engine.sync()
    np.savez(os.path.join(out_dir, "g%d.npz" % r), sums=sums, counts=counts, refs=al.refs.cpu().numpy(),
             params=al.params(), lo=lo, hi=hi)
    al.close()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("backend", ["nccl", "gloo"])
def test_two_ranks_class_sum_exchange(tmp_path, backend):
    """backend nccl = RCCL over xGMI (needs two GPUs); gloo = the same two ranks rehearsed on one GPU, the collective
    staged through host memory"""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU node)")
    from cryo_ralib_amd import synth
    from cryo_ralib_amd.mref import MrefAligner
    nx, ou, nref, xr, n = 90, 36, 6, 3, 400
    port = 29700 + (os.getpid() % 200) + (0 if backend == "nccl" else 200)
    mp.spawn(_worker, args=(2, port, str(tmp_path), nx, ou, nref, xr, n, backend), nprocs=2, join=True)
    a, b = np.load(tmp_path / "g0.npz"), np.load(tmp_path / "g1.npz")
    np.testing.assert_array_equal(a["sums"], b["sums"])
    np.testing.assert_array_equal(a["counts"], b["counts"])
    np.testing.assert_array_equal(a["refs"], b["refs"])
    assert a["counts"].sum() == n
    # one rank over the whole stack: identical assignments, sums equal up to the order of the float additions
    refs = synth.make_references(nref, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 0.5)
    al = MrefAligner(parts, refs, ou, xr, xr, 1.0, preprocess=True)
    al.search()
    al.buf.all_reduce()
    one = al.params()
    both = np.concatenate([a["params"], b["params"]])
    for f in ("ref_id", "mirror", "angle_bin", "shift_idx"):
        np.testing.assert_array_equal(one[f], both[f])
    np.testing.assert_array_equal(al.buf.counts_i.cpu().numpy(), a["counts"])
    s1 = al.buf.sums.cpu().numpy()
    assert np.abs(s1 - a["sums"]).max() <= 4e-6 * np.abs(s1).max()       # n float additions re-associated once
    al.close()
