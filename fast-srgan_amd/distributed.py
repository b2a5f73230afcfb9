"""Data-parallel plumbing: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference has no distributed code; batch sharding is new.  Every op of the hot path is
per-sample (InstanceNorm has no cross-batch statistics, both losses are batch means), so averaging the
per-rank gradients reproduces the single-process gradient at the global batch.  Per optimizer there is ONE
collective: an all-reduce(SUM) of the flat gradient arena (optim.ArenaAdamW) -- 18.7 MB for the
discriminator, 3.7 MB for the generator -- whose 1/world_size is folded into the AdamW kernel.  Messages
this small are latency-bound on the 7 x 153 GB/s xGMI links, so nothing is bucketed or chunked.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun's contract).
    Returns (rank, world_size, local_rank).  A single process (no env) stays un-initialised."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FSR_FORCE_DIST=1: initialise the process group even for ONE process, so that the RCCL path (all-reduce of the gradient
    # arenas, phase graphs around it) can be exercised on a single-GPU box
    force = os.environ.get("FSR_FORCE_DIST", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_distributed():
    """True when gradients go through a collective (a process group exists, whatever its size)."""
    return dist.is_available() and dist.is_initialized()


class GradSync:
    """Gradient exchange for one optimizer's flat gradient buffer.

    `GradSync.record = []` (bench.py, N > 1): every exchange appends (tag, start event, end event) -- the start recorded on
    the current stream when the all-reduce is enqueued, the end after the current stream has been made to wait for it -- so
    the time a step's main stream spends between handing its gradients to RCCL and being allowed to continue can be reported
    per step (`ms_in_allreduce`).  None (the default) records nothing."""

    record = None

    def __init__(self, optimizer, tag=""):
        self.optimizer = optimizer
        self.work = None
        self.tag = tag
        self._ev0 = None
        optimizer.grad_scale = 1.0 / world_size()

    def start(self):
        """Launch the all-reduce (asynchronously where the backend allows); call wait() before step()."""
        if is_distributed():
            if GradSync.record is not None and self.optimizer.flat_grad.is_cuda:
                self._ev0 = torch.cuda.Event(enable_timing=True)
                self._ev0.record()
            self.work = dist.all_reduce(self.optimizer.flat_grad, op=dist.ReduceOp.SUM, async_op=True)

    def run(self):
        self.start()
        self.wait()

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
            if self._ev0 is not None:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record()
                if GradSync.record is not None:
                    GradSync.record.append((self.tag, self._ev0, ev1))
                self._ev0 = None


def all_ranks_ok(ok, device=None):
    """True iff `ok` holds on EVERY rank (one MIN all-reduce of a flag; trivially `ok` in a single process)."""
    if not is_distributed():
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if device is not None and str(device) != "cpu" else None)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def broadcast_parameters(optimizer, src=0):
    """All ranks start from rank `src`'s parameters (one broadcast of the parameter arena)."""
    if is_distributed():
        dist.broadcast(optimizer.flat_param, src=src)
