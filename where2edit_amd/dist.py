"""Data parallelism for the mapper step: one process per GPU, the latent batch sharded by rank, ONE RCCL
all-reduce per step over a flat fp32 bucket of the mapper gradients (LevelsMapper: 3.15 M floats =
12.6 MB), then the identical optimizer step on every rank.  No DDP wrapper, no per-tensor hooks: the
mapper's backward is the last thing autograd computes, so there is nothing to overlap the collective
with, and a single large message suits xGMI's point-to-point links better than 24 small ones.

(The reference trains the StyleCLIP mapper on one GPU -- coach.py:25 hard-codes cuda:0; its only
multi-GPU code is DDP in attention/run_attention.py:1025.)"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, timeout_s=None, force_group=False, use_gpu=True):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).
    Returns (rank, world_size, local_rank).  World size 1 needs no process group.  `timeout_s`: the process group's
    collective timeout (a benchmark wants a rank that lost its partners to fail in minutes, not in the default 10-30).
    `force_group`: create the group at world size 1 too (tools/dp_rccl_selftest.py: the RCCL code path on a one-GPU box).
    `use_gpu=False`: a host-only rendezvous (bench.py --rehearse): no call that initialises the GPU -- a GPU box admits only a few
    processes on its card, and an 8-rank rehearsal must not be 8 of them."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_group) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if (use_gpu and torch.cuda.is_available()) else "gloo"  # "nccl" IS RCCL on ROCm
        kw = {}
        if use_gpu and torch.cuda.is_available():
            dev = local % torch.cuda.device_count()
            torch.cuda.set_device(dev)
            if backend == "nccl":  # bind the communicator to this rank's GPU at creation (RCCL otherwise guesses it from the
                kw["device_id"] = torch.device("cuda", dev)  # rank at the first collective and warns)
        if timeout_s is not None:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=timeout_s)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    elif use_gpu and torch.cuda.is_available():
        # every kernel of libw2e.so launches on the CURRENT device's stream: make the rank's GPU current whatever the
        # backend and world size (gloo self-tests may map several ranks onto one GPU)
        torch.cuda.set_device(local % torch.cuda.device_count())
    return rank, world, local


def shard(latents, rank, world):
    """Contiguous split of the global batch: w[rank*B/R : (rank+1)*B/R]."""
    n = latents.shape[0]
    if n % world != 0:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    per = n // world
    return latents[rank * per:(rank + 1) * per]


class GradBucket:
    """Flat fp32 buffer whose slices ARE the parameters' .grad tensors, so the all-reduce needs no
    gather/scatter copies.  Call `zero()` instead of optimizer.zero_grad() and `all_reduce_mean()`
    between backward and optimizer.step()."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    def zero(self):
        self.flat.zero_()
        off = 0
        for p in self.params:  # re-attach if something replaced .grad (set_to_none)
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def all_reduce_mean(self):
        """grad_global = mean_r grad_r: the loss terms are per-sample means over equal shards."""
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(dist.get_world_size(self.group))
