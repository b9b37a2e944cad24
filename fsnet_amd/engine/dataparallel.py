"""Data parallelism, one process per GPU over RCCL (torch.distributed backend 'nccl' on ROCm).

What the reference gets implicitly from SyncBatchNorm + DistributedDataParallel
(scripts/train.py:100-102), restated explicitly for the HIP engine:
  * BatchNorm batch statistics / backward sums: all-reduce(SUM) of the small f64 buffers between the
    two kernels that produce and consume them (global-batch statistics, count x world).
  * parameter gradients: each network's slice of the flat gradient arena is all-reduced (SUM, async)
    as soon as that network's backward has finished, overlapping the remaining backward; the
    1/world average is folded into the optimizer kernel (grad_scale).
  * rank 0's buffers (BN running stats, depth_bins) are broadcast once at start (DDP's
    broadcast_buffers) — afterwards every rank computes identical running statistics.
"""
import torch
import torch.distributed as dist


class DataParallelContext:
    def __init__(self, meta_arch, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.handles = []
        self.meta = meta_arch
        self.bucket_of = {}
        self._synced = False
        # the 100 small SyncBN exchanges per step go straight to the process group object: the checks and logging
        # of the dist.all_reduce wrapper cost more host time than the exchange itself
        self._pg = group if group is not None else dist.distributed_c10d._get_default_group()
        self._sum = dist.AllreduceOptions()
        self._sum.reduceOp = dist.ReduceOp.SUM
        # ... and, over RCCL, straight to ncclAllReduce on the current stream (rccl_direct.py; None: torch.distributed)
        from .rccl_direct import DirectComm
        self._direct = DirectComm.create(group)

    # ---- small latency-bound exchanges (BN) ------------------------------------------------
    def allreduce_small(self, t, out=None):
        """SUM over the ranks, in place or into `out` (t then keeps the local values)"""
        if self._direct is not None and not torch.cuda.is_current_stream_capturing():
            self._direct.all_reduce_sum(t, out)
            return
        if out is not None:
            out.copy_(t)
            t = out
        self._pg.allreduce([t], self._sum).wait()

    # ---- gradient buckets --------------------------------------------------------------------
    def begin_step(self, meta_arch):
        if not self._synced:
            with torch.no_grad():
                arena = meta_arch._arena
                dist.broadcast(arena.data, src=0, group=self.group)
                for b in meta_arch.buffers():
                    dist.broadcast(b, src=0, group=self.group)
            self._synced = True
        self.handles = []

    def grads_ready(self, module):
        """called by a network's autograd Function when its last pending backward has finished."""
        arena = self.meta._arena
        ent = self.bucket_of.get(id(module))         # walking module.parameters() costs ~0.2 ms per call
        if ent is None or ent[0] is not arena:
            params = [p for p in module.parameters()]
            ent = self.bucket_of[id(module)] = (arena, arena.slice_of(params) if params else None)
        if ent[1] is None:
            return
        from .nets import flush_deferred, join_companions
        if self.world > 1 and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            # under capture join_companions() leaves the companion joins to the end of the step (ROCm 7.2 nested
            # fork-join crash), so this all-reduce node would not depend on the weight-gradient kernels; harmless at
            # world size 1 (the experiment FSNET_AMD_GRAPH_DP=1 covers), wrong beyond it -> the hook falls back to eager
            raise RuntimeError("graph-captured data-parallel steps are not validated beyond world size 1")
        flush_deferred()          # weight-gradient kernels handed to companion streams must have landed in the
        join_companions()         # arena slice before it is reduced
        lo, hi = ent[1]
        h = dist.all_reduce(arena.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.handles.append(h)

    def close(self):
        """release the direct RCCL communicator (before the process group is destroyed)"""
        if self._direct is not None:
            self._direct.close()
            self._direct = None

    def finish(self):
        """wait for the outstanding gradient all-reduces; returns the scale that turns SUM into MEAN."""
        for h in self.handles:
            h.wait()
        self.handles = []
        return 1.0 / self.world
