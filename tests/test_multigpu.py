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


def _worker_loop(rank, world, port, out_dir, nx, ou, xr, n):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from cryo_ralib_amd import dist as rdist
    from cryo_ralib_amd.mref import MrefAligner
    r, local, w = rdist.init_from_env("gloo")
    parts, refs2 = _vanishing_case(nx, ou, xr, n)
    lo, hi = rdist.shard_range(n, w, r)
    al = MrefAligner(parts[lo:hi], refs2, ou, xr, xr, 1.0, device=0, index0=lo, total_nima=n, preprocess=True, myid=r, main_node=0)
    out = {"lo": lo, "hi": hi}
    for it in range(2):
        counts = al.iterate()
        al.engine.sync()
        out["counts%d" % it] = counts.copy()
        out["sums%d" % it] = al.buf.sums.cpu().numpy().copy()
        out["refs%d" % it] = al.refs.cpu().numpy().copy()
        out["params%d" % it] = al.params().copy()
    np.savez(os.path.join(out_dir, "l%d.npz" % r), **out)
    al.close()
    torch.distributed.destroy_process_group()


def _vanishing_case(nx, ou, xr, n):
    """two of four references anti-correlated with every particle: their classes stay empty and are re-seeded"""
    from cryo_ralib_amd import synth
    refs = synth.make_references(4, nx, ou)
    parts, _ = synth.make_particles(refs[:2], n, xr, xr, 0.25, ou=ou)
    refs2 = refs.copy()
    refs2[2] = -refs[0]; refs2[3] = -refs[1]
    return parts, refs2


@pytest.mark.timeout(600)
def test_three_ranks_uneven_shards_and_reseed_broadcast(tmp_path):
    """three ranks rehearsed on one GPU (gloo, collectives staged through host memory) over 61 particles -- shards of 20, 21
    and 20 (MPI_start_end) -- through two iterations of the host driver in which two classes vanish: the even / odd split
    follows the GLOBAL particle index (index0), so sums and counts equal the one-rank run's; the re-seeded references are the
    main node's draw (random.seed(rand_seed); randint(0, nima - 1) over ITS shard, test_mref_gpu_align.py:523-528), broadcast
    to every rank; all ranks hold identical references afterwards"""
    import random
    from cryo_ralib_amd import dist as rdist
    from cryo_ralib_amd.mref import MrefAligner
    nx, ou, xr, n, world = 32, 12, 2, 61, 3
    port = 30100 + (os.getpid() % 200)
    mp.spawn(_worker_loop, args=(world, port, str(tmp_path), nx, ou, xr, n), nprocs=world, join=True)
    outs = [np.load(tmp_path / ("l%d.npz" % r)) for r in range(world)]
    assert [(int(o["lo"]), int(o["hi"])) for o in outs] == [rdist.shard_range(n, world, r) for r in range(world)]
    assert [int(o["hi"]) - int(o["lo"]) for o in outs] == [20, 21, 20]
    for it in range(2):
        for o in outs[1:]:
            np.testing.assert_array_equal(o["sums%d" % it], outs[0]["sums%d" % it])
            np.testing.assert_array_equal(o["counts%d" % it], outs[0]["counts%d" % it])
            np.testing.assert_array_equal(o["refs%d" % it], outs[0]["refs%d" % it])
        assert outs[0]["counts%d" % it].sum() == n
    # iteration 0 against one rank over the whole stack
    parts, refs2 = _vanishing_case(nx, ou, xr, n)
    al = MrefAligner(parts, refs2, ou, xr, xr, 1.0, preprocess=True)
    al.search()
    al.buf.all_reduce()
    one = al.params()
    both = np.concatenate([o["params0"] for o in outs])
    for f in ("ref_id", "mirror", "angle_bin", "shift_idx"):
        np.testing.assert_array_equal(one[f], both[f])
    counts = al.buf.counts_i.cpu().numpy()
    np.testing.assert_array_equal(counts, outs[0]["counts0"])
    assert (counts[2:] < 4).all() and (counts[:2] >= 4).all()
    s1 = al.buf.sums.cpu().numpy()
    assert np.abs(s1 - outs[0]["sums0"]).max() <= 4e-6 * np.abs(s1).max()       # per (class, parity): the same images, re-associated
    # the re-seeded classes: the main node's particles k0, k1 of its own shard (20 particles), normalised under the mask
    rng = random.Random(1000)
    lo0, hi0 = rdist.shard_range(n, world, 0)
    want = al.refs.clone()
    for j in (2, 3):
        want[j] = al.particles[lo0 + rng.randint(0, hi0 - lo0 - 1)]
    al.refs = want
    al._normalize_refs([2, 3])
    np.testing.assert_allclose(outs[0]["refs0"][2:], al.refs[2:].cpu().numpy(), rtol=0, atol=1e-6)
    al.close()


def _worker_reffree(rank, world, port, out_dir, nx, ou, xr, n):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from cryo_ralib_amd import dist as rdist, synth
    from cryo_ralib_amd.mref import RefFreeAligner
    r, local, w = rdist.init_from_env("gloo")
    refs = synth.make_references(1, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 0.5, ou=ou)
    lo, hi = rdist.shard_range(n, w, r)
    al = RefFreeAligner(parts[lo:hi], ou, xr, xr, 1.0, device=0, index0=lo, total_nima=n)
    raws = []
    for it in range(3):
        al.iterate(-1, "ref_ali2d")
        raws.append(al.raw_avg.cpu().numpy().copy())
    np.savez(os.path.join(out_dir, "f%d.npz" % r), raw=np.stack(raws), tavg=al.tavg.cpu().numpy())
    al.close()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_reference_free_raw_average_is_the_reduced_one(tmp_path):
    """aqc.hdf holds (ave1 + ave2) / total_nima AFTER reduce_EMData_to_root, for every iteration including the sum_oe average
    of the first (test_reffree_gpu_align.py:365-383): two ranks (gloo rehearsal on one GPU) keep the same raw average in
    RefFreeAligner.raw_avg as one rank over the whole stack, on both ranks, in all three iterations"""
    from cryo_ralib_amd import synth
    from cryo_ralib_amd.mref import RefFreeAligner
    nx, ou, xr, n = 32, 12, 2, 75
    port = 30500 + (os.getpid() % 200)
    mp.spawn(_worker_reffree, args=(2, port, str(tmp_path), nx, ou, xr, n), nprocs=2, join=True)
    a, b = np.load(tmp_path / "f0.npz"), np.load(tmp_path / "f1.npz")
    np.testing.assert_array_equal(a["raw"], b["raw"])
    refs = synth.make_references(1, nx, ou)
    parts, _ = synth.make_particles(refs, n, xr, xr, 0.5, ou=ou)
    al = RefFreeAligner(parts, ou, xr, xr, 1.0)
    for it in range(3):
        al.iterate(-1, "ref_ali2d")
        one = al.raw_avg.cpu().numpy()
        assert np.abs(one).max() > 0
        np.testing.assert_allclose(a["raw"][it], one, rtol=0, atol=2e-5 * np.abs(one).max())
    # iteration 0's raw average is the plain mean of the stack
    np.testing.assert_allclose(a["raw"][0], parts.mean(0), rtol=0, atol=1e-5 * np.abs(parts).max())
    al.close()


_RCCL_WORLD1 = r"""
import os, sys
sys.path.insert(0, %(root)r)
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r,
                  HSA_ENABLE_IPC_MODE_LEGACY="0", RALIGN_FORCE_COLLECTIVE="1")
import numpy as np
import torch
import torch.distributed as dist
from cryo_ralib_amd import dist as rdist, synth
from cryo_ralib_amd.mref import MrefAligner
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
nx, ou, nref, xr, n = 90, 36, 6, 3, 200
refs = synth.make_references(nref, nx, ou)
parts, _ = synth.make_particles(refs, n, xr, xr, 0.5)
al = MrefAligner(parts, refs, ou, xr, xr, 1.0, device=0, preprocess=True)
al.search()
al.engine.sync()
before = al.buf.sums.clone()
cnt = al.buf.counts_i.clone()
for mode in ("1", "0"):                      # rank-ordered all-gather + sum, then the plain all-reduce
    os.environ["RALIGN_ORDERED_REDUCE"] = mode
    al.buf.all_reduce()
    torch.cuda.synchronize()
    assert torch.equal(al.buf.sums, before), mode
    assert torch.equal(al.buf.counts_i, cnt) and int(cnt.sum()) == n
t = al.refs.clone()
rdist.broadcast(t, src=0)
torch.cuda.synchronize()
assert torch.equal(t, al.refs)
# the exchange buffers of BASELINE configs[1] (R = 10, 90 x 90: 648 KB -> rank-ordered all-gather + sum) and configs[4] (R = 100,
# 256 x 256: 52.4 MB -> the plain all-reduce) with the DEFAULT size rule, device tensors, bit for bit; and the broadcast of the
# configs[4] reference stack (26 MB)
os.environ.pop("RALIGN_ORDERED_REDUCE", None)
dev = torch.device("cuda", 0)
for nref_b, nx_b, want in ((10, 90, "ordered"), (100, 256, "all_reduce")):
    buf = rdist.ClassSumBuffer(nref_b, nx_b, dev, extra=2)
    g = torch.Generator(device=dev); g.manual_seed(5 + nref_b)
    buf.sums.normal_(generator=g)
    buf.counts_i.copy_(torch.randint(0, 5000, (nref_b,), generator=g, device=dev, dtype=torch.int32))
    buf.extra_f.copy_(torch.tensor([1.25, -3.5], device=dev))
    assert buf.flat.is_cuda and buf.flat.numel() * 4 == (nref_b * 2 * nx_b * nx_b + nref_b + 2) * 4
    ref_s, ref_c, ref_e = buf.sums.clone(), buf.counts_i.clone(), buf.extra_f.clone()
    buf.all_reduce()
    torch.cuda.synchronize()
    assert buf.last_path == want, (nref_b, nx_b, buf.last_path)
    assert torch.equal(buf.sums, ref_s) and torch.equal(buf.counts_i, ref_c) and torch.equal(buf.extra_f, ref_e), (nref_b, nx_b)
    stack = buf.sums[:, 0].contiguous()
    keep = stack.clone()
    rdist.broadcast(stack, src=0)
    torch.cuda.synchronize()
    assert stack.is_cuda and torch.equal(stack, keep)
    del buf, stack, keep, ref_s
x = torch.arange(1024, dtype=torch.float32, device="cuda")
dist.all_reduce(x)
assert float(x[1000]) == 1000.0
assert rdist.max_over_ranks(3.5, torch.device("cuda", 0)) == 3.5
rdist.barrier()
al.close()
dist.destroy_process_group()
print("RCCL world-1 ok")
"""


@pytest.mark.timeout(600)
def test_rccl_executes_in_a_process_group_of_one_rank():
    """RCCL itself on the one-GPU box: a world-size-1 `nccl` process group in a fresh child process (launcher environment set
    before any GPU call) runs ClassSumBuffer.all_reduce in both forms -- the rank-ordered all-gather + sum and the plain
    all-reduce -- dist.broadcast, an all-reduce of a plain tensor, the max-over-ranks of bench.py and a barrier on device
    tensors, so that a missing or mismatched librccl, a communicator that cannot be built or a failing collective kernel shows
    here and not first in the driver's 8-GPU run.  (One rank: sums, gathers and broadcasts return their input.)
    Replaces reduce_EMData_to_root x 2R + mpi_reduce + bcast_EMData_to_all x R (test_mref_gpu_align.py:495-499, 572-575)."""
    import subprocess
    port = str(30500 + (os.getpid() % 200))
    code = _RCCL_WORLD1 % {"root": ROOT, "port": port}
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "RCCL world-1 ok" in out.stdout
