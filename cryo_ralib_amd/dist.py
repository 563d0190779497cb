"""Particle sharding and the one collective of the path.

The reference shards particles contiguously over MPI ranks with MPI_start_end
(test_mref_gpu_align.py:289,1384) and, once per iteration, reduces the 2R class sums and
R counts to rank 0 (reduce_EMData_to_root x 2R + mpi_reduce, :495-499) and broadcasts the
new references back (bcast_EMData_to_all x R, :572-575).  Here that is ONE in-place
all-reduce (RCCL over xGMI when the tensors live in HBM, gloo on CPU tensors in the tests)
of a single flat buffer [R*2*nx*nx sums | R counts | extra]; every rank then performs the
same deterministic reference update, which replaces the broadcast.
"""
import os

import torch
import torch.distributed as dist

from .geometry import mpi_start_end


def init_from_env(backend=None):
    """one process per GPU, launched by torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # RALIGN_DIST_BACKEND=gloo lets several ranks rehearse on one GPU (RCCL refuses duplicate devices)
            backend = os.environ.get("RALIGN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def shard_range(total, world, rank):
    """contiguous particle range of `rank` (MPI_start_end rule)."""
    return mpi_start_end(total, world, rank)


ORDERED_MAX_BYTES = 8 << 20       # gathered size up to which the rank-ordered sum replaces the all-reduce


def _collective_needed():
    """more than one rank -- or RALIGN_FORCE_COLLECTIVE=1 in a process group of ONE rank: the collectives then run as they
    would on many (RCCL's library load, communicator set-up and kernels execute; the result of a sum / gather / broadcast over
    one rank is the input), which is how the one-GPU test box exercises the RCCL path (tests/test_multigpu.py)"""
    if not dist.is_initialized():
        return False
    return dist.get_world_size() > 1 or os.environ.get("RALIGN_FORCE_COLLECTIVE") == "1"


class ClassSumBuffer:
    """flat fp32 buffer [R][2][nx][nx] sums | [R] counts | [extra] scalars, all-reduced in place.

    Counts travel as fp32 inside the same buffer (exact below 2^24 members per class per
    rank-sum, i.e. 16.7 M particles per class) so that one collective suffices."""

    def __init__(self, nref, nx, device, extra=0):
        self.nref, self.nx, self.extra = nref, nx, extra
        self.nsum = nref * 2 * nx * nx
        self.flat = torch.zeros(self.nsum + nref + extra, dtype=torch.float32, device=device)
        self.sums = self.flat[:self.nsum].view(nref, 2, nx, nx)
        self.counts_f = self.flat[self.nsum:self.nsum + nref]
        self.extra_f = self.flat[self.nsum + nref:]
        self.counts_i = torch.zeros(nref, dtype=torch.int32, device=device)
        self._gather = None
        self.last_path = None       # which collective the last all_reduce ran: "ordered" | "all_reduce" | None (one rank, nothing to do)

    def zero_(self):
        self.flat.zero_()
        self.counts_i.zero_()

    def all_reduce(self):
        """sum over ranks; afterwards counts_i holds the global member counts on every rank.

        Small buffers ("ordered", world x bytes <= ORDERED_MAX_BYTES: 648 KB per rank at R = 10 / 90^2): all-gather the
        per-rank buffers and add them in rank order, ((S0 + S1) + S2) + ... -- every rank performs the same additions in
        the same order, so the result is bitwise identical on all ranks and does not depend on which algorithm (ring, tree,
        direct) RCCL picks for the message size or on the xGMI topology.  Larger buffers (3.24 MB per rank at R = 50,
        52.4 MB at R = 100 / 256^2, where the gather would move world x the bytes): one plain RCCL all-reduce (sum) -- still
        the same bits on every rank of a run, but the association of the float additions follows RCCL's algorithm, so
        runs on different rank counts / topologies agree to rounding only.  RALIGN_ORDERED_REDUCE=1 / 0 forces either."""
        self.counts_f.copy_(self.counts_i.to(torch.float32))
        self.last_path = None
        if _collective_needed():
            world = dist.get_world_size()
            # gloo rehearsal of several ranks on one GPU (RALIGN_DIST_BACKEND=gloo): the collective runs on a host copy
            staged = self.flat.is_cuda and dist.get_backend() == "gloo"
            work = self.flat.cpu() if staged else self.flat
            mode = os.environ.get("RALIGN_ORDERED_REDUCE", "")
            ordered = mode == "1" or (mode != "0" and world * work.numel() * 4 <= ORDERED_MAX_BYTES)
            if ordered and world <= 64:
                if self._gather is None or self._gather.shape[0] != world or self._gather.device != work.device:
                    self._gather = torch.empty((world,) + work.shape, dtype=work.dtype, device=work.device)
                dist.all_gather_into_tensor(self._gather.view(-1), work)
                work.copy_(self._gather[0])
                for r in range(1, world):
                    work.add_(self._gather[r])
                self.last_path = "ordered"
            else:
                dist.all_reduce(work, op=dist.ReduceOp.SUM)
                self.last_path = "all_reduce"
            if staged:
                self.flat.copy_(work)
        self.counts_i.copy_(self.counts_f.round().to(torch.int32))
        return self


def broadcast(t, src=0):
    """in-place broadcast of tensor `t` from rank `src` (bcast_EMData_to_all, test_mref_gpu_align.py:572-575); under the gloo
    rehearsal of several ranks on one GPU the collective runs on a host copy, like ClassSumBuffer.all_reduce"""
    if not _collective_needed():
        return t
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src)
    return t


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
