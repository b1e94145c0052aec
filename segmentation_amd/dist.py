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
    def __init__(self, process_group=None, overlap=True, algo=None):
        self.enabled = dist.is_available() and dist.is_initialized()
        self.group = process_group
        self.world = dist.get_world_size(process_group) if self.enabled else 1
        self.rank = dist.get_rank(process_group) if self.enabled else 0
        self.overlap = overlap
        self.pending = []
        # 'allreduce': one RCCL all-reduce per bucket (RCCL picks ring / tree and the protocol from the message size).
        # 'rs_ag': the bucket as reduce-scatter + all-gather over equal 1/world slices -- the two-phase form SURVEY 8(e) asks for on
        # xGMI's point-to-point links (every rank owns 1/world of the bucket, sends the other slices straight to their owners over
        # all 7 links and gathers the reduced slices back: 2 x bucket / world bytes per link instead of 2 x 7/8 x bucket through one
        # link of a ring).  Same sums in the same per-element order on every rank, so replicas stay identical.  SEG_DP_ALGO selects.
        import os
        self.algo = algo or os.environ.get('SEG_DP_ALGO', 'allreduce')
        self._dry = os.environ.get('SEG_DP_DRY', '0') == '1'
        self.force_collectives = os.environ.get('SEG_DP_FORCE', '0') == '1'     # world 1: run the (identity) collectives anyway
        # the step's stream set-up is TUNED for RCCL's company (own auxiliary stream, no high-priority stream, lone-launch filter-gradient
        # targets) only where collectives are really issued: a world-1 run keeps the single-GPU set-up and pays for the bucket markers only
        self.tuned = self.enabled and (self.world > 1 or self.force_collectives)
        self.rs_min = 4096                      # rs_ag: elements per rank below which a bucket goes through one all-reduce
        self._has_rs = None
        if self.algo not in ('allreduce', 'rs_ag'):
            raise ValueError("SEG_DP_ALGO must be 'allreduce' or 'rs_ag'")

    def _backend_has_reduce_scatter(self):
        """RCCL (backend "nccl") has reduce_scatter_tensor / all_gather_into_tensor; gloo has neither reduce-scatter form.  There the
        reduce-scatter is EMULATED -- an all-reduce of a copy of the bucket body, of which this rank keeps only its own slice, so
        that the slice arithmetic, the all-gather into the in-place views and the tail all-reduce below are the code RCCL runs."""
        if self._has_rs is None:
            try:
                self._has_rs = dist.get_backend(self.group) == 'nccl'
            except Exception:                              # noqa
                self._has_rs = False
        return self._has_rs

    def all_reduce_bucket(self, flat, lo, hi):
        if not self.enabled or hi <= lo:
            return
        if self.world == 1 and not self.force_collectives:
            # A one-rank "all-reduce" is the identity, but RCCL still launches its copy kernels for it: measured 2 % of the C2 step
            # (profiles/r03_dp_overhead.txt), overhead an 8-rank run does not have in that form.  bench.py --force-dist keeps them
            # as a diagnostic of the data-parallel step structure (force_collectives / SEG_DP_FORCE=1).
            return
        if self._dry:                           # (diagnostic: the step's own data-parallel overhead without RCCL's kernels)
            return
        n, w_ = hi - lo, self.world
        if self.algo == 'rs_ag' and w_ > 1 and n >= self.rs_min * w_:
            # equal slices (the tail that does not divide goes through a small all-reduce of its own)
            per = n // w_
            body = flat[lo:lo + per * w_]
            mine = body[self.rank * per:(self.rank + 1) * per]
            works = []
            if self._backend_has_reduce_scatter():
                # both collectives run in order on the process group's one RCCL stream: no host- or stream-level wait between them
                # (a wait here would park the issuing side stream behind the reduce-scatter)
                works.append(dist.reduce_scatter_tensor(mine, body, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                tmp = body.clone()
                dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
                mine.copy_(tmp[self.rank * per:(self.rank + 1) * per])     # the other slices of `body` keep this rank's local sums
            works.append(dist.all_gather_into_tensor(body, mine, group=self.group, async_op=True))
            if per * w_ < n:
                works.append(dist.all_reduce(flat[lo + per * w_:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            works = [dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        for w in works:
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
