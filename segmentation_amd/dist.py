"""Data-parallel gradient exchange: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code.  U-Net/FCN have no BatchNorm, so a global batch split
N ways equals one device processing the whole batch up to summation order: every rank takes the mean
loss over its own pixels, gradients are SUM-all-reduced and Adam scales them by 1/world.  The gradients
live in ONE flat fp32 arena laid out in backward-production order, so a bucket is a contiguous slice
[lo, hi) that is complete as soon as the backward segment producing it has been enqueued; its all-reduce
runs on RCCL's own stream while the next segment's dgrad/wgrad kernels run on the compute stream."""
import torch
import torch.distributed as dist


class DataParallel(object):
    def __init__(self, process_group=None, overlap=True):
        self.enabled = dist.is_available() and dist.is_initialized()
        self.group = process_group
        self.world = dist.get_world_size(process_group) if self.enabled else 1
        self.rank = dist.get_rank(process_group) if self.enabled else 0
        self.overlap = overlap
        self.pending = []

    def all_reduce_bucket(self, flat, lo, hi):
        if not self.enabled or hi <= lo:        # world 1 still runs the collective when a group exists (1-GPU RCCL path)
            return
        w = dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if self.overlap:
            self.pending.append(w)
        else:
            w.wait()

    def wait_all(self):
        for w in self.pending:
            w.wait()           # stream-level wait on CUDA/HIP tensors; blocking on gloo/CPU
        self.pending = []

    def broadcast_(self, flat, src=0):
        if self.enabled and self.world > 1:
            dist.broadcast(flat, src=src, group=self.group)


def shard_batch(global_batch, world, rank):
    """Contiguous, equal slices of the minibatch."""
    if global_batch % world:
        raise ValueError('global batch %d not divisible by world size %d' % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per
