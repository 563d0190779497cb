"""world_size-2 gloo test of the N>1 host path: contiguous sharding, the single all-reduce
of [sums | counts | extra] and the max-over-ranks timing reduction."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, nref, nx, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from cryo_ralib_amd import dist as rdist
    r, _, w = rdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    lo, hi = rdist.shard_range(total, world, rank)
    g = torch.Generator().manual_seed(1234)
    imgs = torch.randn((total, nx, nx), generator=g)
    cls = torch.randint(0, nref, (total,), generator=g)
    buf = rdist.ClassSumBuffer(nref, nx, torch.device("cpu"), extra=2)
    buf.zero_()
    for i in range(lo, hi):
        buf.sums[cls[i], i % 2] += imgs[i]
        buf.counts_i[cls[i]] += 1
    buf.extra_f[0] = float(hi - lo)
    buf.all_reduce()
    t = rdist.max_over_ranks(1.0 + rank, torch.device("cpu"))
    s = rdist.sum_over_ranks(1.0, torch.device("cpu"))
    rdist.barrier()
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), sums=buf.sums.numpy(), counts=buf.counts_i.numpy(),
             extra=buf.extra_f.numpy(), lo=lo, hi=hi, t=t, s=s)
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_class_sum_allreduce_world2(tmp_path):
    world, total, nref, nx = 2, 37, 3, 8
    port = 29500 + (os.getpid() % 400)
    mp.spawn(_worker, args=(world, port, total, nref, nx, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / ("r%d.npz" % r)) for r in range(world)]
    g = torch.Generator().manual_seed(1234)
    imgs = torch.randn((total, nx, nx), generator=g)
    cls = torch.randint(0, nref, (total,), generator=g)
    want = torch.zeros((nref, 2, nx, nx)); cnt = np.zeros(nref, np.int32)
    for i in range(total):
        want[cls[i], i % 2] += imgs[i]
        cnt[cls[i]] += 1
    assert outs[0]["hi"] == outs[1]["lo"] and outs[0]["lo"] == 0 and outs[1]["hi"] == total
    for o in outs:
        np.testing.assert_allclose(o["sums"], want.numpy(), atol=1e-5)
        np.testing.assert_array_equal(o["counts"], cnt)
        assert o["extra"][0] == total and o["t"] == 2.0 and o["s"] == 2.0
    np.testing.assert_array_equal(outs[0]["sums"], outs[1]["sums"])      # every rank holds identical sums


def _worker_ordered(rank, world, port, out_dir, ordered):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), RALIGN_ORDERED_REDUCE="1" if ordered else "0")
    from cryo_ralib_amd import dist as rdist
    rdist.init_from_env("gloo")
    buf = rdist.ClassSumBuffer(2, 6, torch.device("cpu"))
    g = torch.Generator().manual_seed(77 + rank)
    buf.flat[:buf.nsum] = torch.randn(buf.nsum, generator=g) * (10.0 ** rank)      # magnitudes that make the order matter
    buf.counts_i[:] = torch.tensor([rank + 1, 2 * rank])
    buf.all_reduce()
    np.savez(os.path.join(out_dir, "o%d_%d.npz" % (int(ordered), rank)), flat=buf.flat.numpy(), counts=buf.counts_i.numpy())
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
def test_ordered_reduction_is_rank_order_sum_on_every_rank(tmp_path):
    """the default cross-rank reduction adds the per-rank class sums in rank order, ((S0 + S1) + S2): bitwise the
    same on every rank and independent of the collective's algorithm"""
    world = 3
    port = 29900 + (os.getpid() % 400)
    mp.spawn(_worker_ordered, args=(world, port, str(tmp_path), True), nprocs=world, join=True)
    parts = []
    for r in range(world):
        g = torch.Generator().manual_seed(77 + r)
        parts.append(torch.randn(2 * 2 * 6 * 6, generator=g) * (10.0 ** r))
    want = ((parts[0] + parts[1]) + parts[2]).numpy()
    for r in range(world):
        o = np.load(tmp_path / ("o1_%d.npz" % r))
        np.testing.assert_array_equal(o["flat"][:want.size], want)
        np.testing.assert_array_equal(o["counts"], [6, 6])


def _worker_threshold(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("RALIGN_ORDERED_REDUCE", None)
    from cryo_ralib_amd import dist as rdist
    rdist.init_from_env("gloo")
    calls = []
    real_gather, real_reduce = rdist.dist.all_gather_into_tensor, rdist.dist.all_reduce
    rdist.dist.all_gather_into_tensor = lambda *a, **k: (calls.append("gather"), real_gather(*a, **k))[1]
    rdist.dist.all_reduce = lambda *a, **k: (calls.append("reduce"), real_reduce(*a, **k))[1]
    res = {}
    for name, limit in (("small", rdist.ORDERED_MAX_BYTES), ("large", 1024)):
        rdist.ORDERED_MAX_BYTES = limit
        buf = rdist.ClassSumBuffer(3, 8, torch.device("cpu"))
        buf.flat[:buf.nsum] = float(rank + 1)
        buf.counts_i[:] = rank + 2
        del calls[:]
        buf.all_reduce()
        res[name] = (list(calls), float(buf.flat[0]), int(buf.counts_i[0]))
        assert buf.last_path == ("ordered" if name == "small" else "all_reduce"), (name, buf.last_path)       # what the RCCL test reads
    np.save(os.path.join(out_dir, "t%d.npy" % rank), np.array([res["small"][1], res["small"][2], res["large"][1], res["large"][2],
                                                                 res["small"][0] == ["gather"], res["large"][0] == ["reduce"]], float))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
def test_large_buffers_take_the_plain_all_reduce(tmp_path):
    """the rank-ordered gather + sum serves buffers up to ORDERED_MAX_BYTES (gathered size); above it -- nref = 50 and the
    256 x 256 / nref = 100 case -- the one collective of the path is a plain all-reduce (sum), as north_star names it"""
    world = 2
    port = 30300 + (os.getpid() % 400)
    mp.spawn(_worker_threshold, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        v = np.load(tmp_path / ("t%d.npy" % r))
        assert list(v) == [3.0, 5.0, 3.0, 5.0, 1.0, 1.0]
