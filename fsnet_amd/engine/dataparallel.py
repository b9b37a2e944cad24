"""Data parallelism, one process per GPU over RCCL (torch.distributed backend 'nccl' on ROCm).

What the reference gets implicitly from SyncBatchNorm + DistributedDataParallel
(scripts/train.py:100-102), restated explicitly for the HIP engine:
  * BatchNorm batch statistics / backward sums: all-reduce(SUM) of the small f64 buffers between the
    two kernels that produce and consume them (global-batch statistics, count x world).
  * parameter gradients: slices of the flat gradient arena are all-reduced (SUM) as soon as the weight gradients of
    a ResNet stage / decoder have been issued — buckets in reverse parameter order like DDP's, on a communication
    stream beside the rest of the backward; the 1/world average is folded into the optimizer kernel (grad_scale).
  * rank 0's parameters and buffers (BN running stats, depth_bins) are broadcast once at start (DDP's
    broadcast) — afterwards every rank computes identical updates.

Transport.  With the nccl backend every collective of a step — the ~100 small SyncBN exchanges and the gradient
buckets — goes through ONE direct RCCL communicator (rccl_direct.py: ncclAllReduce on the engine's own HIP
streams).  torch.distributed's communicator is then only used at set-up and for barriers, never concurrently with
the direct one (two communicators executing collectives on one device at the same time can deadlock when the
device-side order differs across ranks).  No c10d work objects exist during a step, so the step can be captured
into a hipGraph: RCCL's launches become graph nodes, RCCL itself orders the launches of one communicator across the
chain streams — in issue order, which at N > 1 couples the depth and the pose chain wherever they exchange.
With gloo (CPU tests, shared-device rigs), or if the direct path fails its start-up self-test, the
collectives fall back to torch.distributed and the step is issued eagerly.
"""
import os

import torch
import torch.distributed as dist


class DataParallelContext:
    def __init__(self, meta_arch, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.handles = []
        self.meta = meta_arch
        self._slices = {}          # id(module) -> (arena, (lo, hi) or None)
        self._done = {}            # id(module) -> [(lo, hi)] reduced in this step
        self._expected = {}        # id(module) -> module: ran a training forward in this step
        self._synced = False
        self.n_small = self.n_bucket = 0   # collectives issued in the current step (bench.py reports them at N > 1)
        self.encoder_pass_ms = None        # the training hook's autotune result (two chains / two lanes), once it has run
        self._pg = group if group is not None else dist.distributed_c10d._get_default_group()
        self._sum = dist.AllreduceOptions()
        self._sum.reduceOp = dist.ReduceOp.SUM
        from .rccl_direct import DirectComm
        self._direct = DirectComm.create(group)
        self._agreement = None
        self._comm_stream = None
        self._graph_owners = []    # weak references to hooks whose captured step contains this communicator's nodes
        self._parked = []          # captured steps (they hold state of this communicator) — destroyed in close(), before it
        self.capturable = False
        # Where the weight gradients run under data parallelism (they feed nothing downstream but their gradient bucket):
        #   inline     on their chain, so one event after a stage's last weight gradient covers its bucket;
        #   companion  on the chains' companion streams as in the single-GPU step — a bucket's all-reduce then waits for the
        #              chain and for every companion holding work;
        #   tail       inline, except the depth decoder's, which run at the tail of the pose chain's stream when the backward
        #              has been issued (the depth chain — decoder, then encoder — is the longer one: nets.flush_tail).
        # FSNET_AMD_DP_WGRAD names one; "auto" (default) lets the training hook's autotune time them on the ranks present.
        self.wgrad_env = os.environ.get("FSNET_AMD_DP_WGRAD", "auto").lower()
        assert self.wgrad_env in ("auto", "inline", "companion", "tail"), self.wgrad_env
        self.wgrad_mode = "inline" if self.wgrad_env == "auto" else self.wgrad_env
        if self._direct is not None:
            from .runtime import RT
            self._comm_stream = RT.new_stream(self._direct.device)
            self.capturable = self._direct.capture_ok

    @property
    def direct(self):
        return self._direct is not None

    @property
    def wgrad_companions(self):
        return self.wgrad_mode == "companion"

    @property
    def decoder_tail(self):
        return self.wgrad_mode == "tail"

    # ---- small latency-bound exchanges (BN) ------------------------------------------------
    def allreduce_small(self, t, out=None):
        """SUM over the ranks, in place or into `out` (t then keeps the local values)"""
        self.n_small += 1
        if self._direct is not None:
            self._direct.all_reduce_sum(t, out)          # current stream; a graph node under capture
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
                if self._direct is not None:
                    # (on the direct communicator: the engine creates no torch.distributed NCCL work object, ever)
                    self._direct.broadcast(arena.data, 0)
                    for b in meta_arch.buffers():
                        if not b.is_cuda:
                            raise RuntimeError("data parallel: buffer on %s — the direct communicator broadcasts device "
                                               "memory only (move the module to the GPU first)" % b.device)
                        if b.is_contiguous():
                            self._direct.broadcast(b, 0)
                        else:
                            tmp = b.contiguous()
                            self._direct.broadcast(tmp, 0)
                            b.copy_(tmp)
                else:
                    dist.broadcast(arena.data, src=0, group=self.group)
                    for b in meta_arch.buffers():
                        dist.broadcast(b, src=0, group=self.group)
            from .runtime import RT
            RT.bump_weights()        # packed MFMA operands follow the broadcast weights
            torch.cuda.synchronize() if torch.cuda.is_available() else None
            self._synced = True
        # a forward that never met its backward (validation-style call, caught exception) must not leave a network
        # waiting forever: the pending counters restart with every step
        for m in meta_arch.modules():
            if "_pending" in m.__dict__ and isinstance(m.__dict__["_pending"], int):
                m._pending = 0
        self.handles = []
        self._done = {}
        self._expected = {}
        self.n_small = self.n_bucket = 0
        # bias sums a backward that raised left behind would be flushed into this step's freshly zeroed gradients
        from .nets import drop_inline_bias
        drop_inline_bias()

    def note_forward(self, module):
        """a network ran a training forward: its gradients must be reduced before finish()"""
        self._expected[id(module)] = module

    def _slice(self, module, arena):
        ent = self._slices.get(id(module))         # walking module.parameters() costs ~0.2 ms per call
        if ent is None or ent[0] is not arena:
            params = [p for p in module.parameters() if arena.owns(p)]
            ent = self._slices[id(module)] = (arena, arena.slice_of(params) if params else None)
        return ent[1]

    def _reduce_range(self, arena, lo, hi):
        if arena.grad.is_cuda:
            from .nets import flush_inline_bias
            flush_inline_bias(torch.cuda.current_stream(arena.grad.device))   # bias sums collected along this chain
        if hi <= lo:
            return
        self.n_bucket += 1
        g = arena.grad[lo:hi]
        if self._direct is not None:
            # the bucket's weight gradients were issued on the current (chain) stream: the communication stream
            # picks up from there, the chain goes on with the rest of the backward
            from .nets import flush_deferred, late_call, pending_companions
            cur = torch.cuda.current_stream(g.device)
            comm, companions = self._comm_stream, self.wgrad_companions

            def reduce_bucket(ev=None):
                if ev is None:
                    comm.wait_stream(cur)
                else:
                    comm.wait_event(ev)
                if companions:
                    for ws in pending_companions():
                        comm.wait_stream(ws)
                with torch.cuda.stream(comm):
                    self._direct.all_reduce_sum(g)
            if companions:
                flush_deferred(cur)                  # what the chain still holds goes to its companion (inside a capture:
            #                                          at the end of the pass, the bucket behind it — nets.flush_deferred)
            if not late_call(cur, reduce_bucket):
                if companions:
                    flush_deferred(cur, now=True)
                reduce_bucket()
            return
        from .nets import flush_deferred, join_companions
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("torch.distributed gradient buckets cannot be captured into a hipGraph")
        flush_deferred(now=True)  # weight-gradient kernels handed to companion streams must have landed in the
        join_companions()         # arena slice before it is reduced
        self.handles.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def partial_ready(self, module, sub_modules):
        """the weight gradients of `sub_modules` (children of `module`, e.g. one ResNet stage) are issued: reduce
        their slice now.  grads_ready(module) later reduces what is left."""
        arena = self.meta._arena
        sl = self._slice(module, arena)
        if sl is None:
            return
        params = [p for m in sub_modules for p in m.parameters() if arena.owns(p)]
        if not params:
            return
        lo, hi = arena.slice_of(params)
        assert sl[0] <= lo and hi <= sl[1]
        if any(a < hi and lo < b for a, b in self._done.get(id(module), [])):
            raise RuntimeError("gradient range [%d, %d) of %s was already reduced in this step" %
                               (lo, hi, type(module).__name__))
        self._done.setdefault(id(module), []).append((lo, hi))
        self._reduce_range(arena, lo, hi)

    def grads_ready(self, module):
        """called by a network's autograd Function when its last pending backward has finished: reduces the part of
        the network's arena slice that partial_ready() has not covered yet."""
        arena = self.meta._arena
        sl = self._slice(module, arena)
        if sl is None:
            return
        done = sorted(self._done.get(id(module), []))
        pos = sl[0]
        for lo, hi in done:
            self._reduce_range(arena, pos, lo)
            pos = max(pos, hi)
        self._reduce_range(arena, pos, sl[1])
        self._done[id(module)] = [(sl[0], sl[1])]

    def all_agree(self, ok):
        """True on every rank iff `ok` on every rank.  Used after a rank-local decision that changes WHICH collectives
        a rank will issue — a rank that fell back to eager launches while the others replay a captured step would hang
        them.  Through the process group's store (rccl_direct.StoreAgreement): no collective, nothing for the NCCL
        watchdog to poll, valid for any backend."""
        if self._agreement is None:
            from .rccl_direct import StoreAgreement
            own = self._direct.agreement if self._direct is not None else None
            self._agreement = own if own is not None else StoreAgreement(self.group)
        return self._agreement.all_agree(ok)

    def gather_floats(self, values):
        """[[rank 0's values], [rank 1's], ...] on every rank, through the store (no collective: callable between steps
        whatever the transport)"""
        import struct
        if self._agreement is None:
            self.all_agree(True)          # (creates the agreement object; every rank takes this branch together)
        blobs = self._agreement.gather(struct.pack("<%dd" % len(values), *[float(v) for v in values]))
        return [list(struct.unpack("<%dd" % len(values), b)) for b in blobs]

    def reset_direct(self):
        """after a capture that failed on some rank: the direct communicator may hold half-captured launches of the rank
        whose capture died — every rank closes it and opens a fresh one (collective; falls back to torch.distributed if
        that fails) before the eager step reuses it"""
        if self._direct is None:
            return
        # a rank whose own capture succeeded still holds a hipGraph with this communicator's RCCL nodes: it goes first (a
        # graph that outlives its communicator aborts the process when it is finally freed — see close())
        for ref in self._graph_owners:
            owner = ref()
            if owner is not None:
                owner.reset_graph()
        self._graph_owners = []
        self._destroy_parked()
        self._direct.close()
        from .rccl_direct import DirectComm
        self._agreement = None
        self._direct = DirectComm.create(self.group)
        self.capturable = False                      # this context steps eagerly from here on

    def park_graph(self, graph):
        """a captured step with this communicator's RCCL nodes: kept alive (BaseTrainingHook: graph execs are not destroyed
        while the process goes on capturing) until close() destroys it, before the communicator"""
        self._parked.append(graph)

    def _destroy_parked(self):
        for g in self._parked:
            try:
                g.reset()
            except Exception:       # noqa: BLE001 — a half-captured graph of a failed capture
                pass
        self._parked = []

    def note_graph_owner(self, owner):
        """`owner.reset_graph()` drops a hipGraph captured with this context's collectives"""
        import weakref
        self._graph_owners.append(weakref.ref(owner))

    def close(self):
        """release the direct RCCL communicator (before the process group is destroyed).  Captured steps that contain
        its collectives are destroyed first: RCCL keeps per-communicator state for graph-captured launches and a
        graph outliving its communicator aborts the process when it is finally freed."""
        for ref in self._graph_owners:
            owner = ref()
            if owner is not None:
                owner.reset_graph()
        self._graph_owners = []
        self._destroy_parked()
        for c in ([self._direct] if self._direct is not None else []):
            c.close()
        self._direct = None
        self._agreement = None
        from .runtime import RT
        if RT.dp is self:
            RT.override_lanes(None)          # the autotune's choice belonged to this context
        if self._comm_stream is not None:
            from .runtime import RT
            RT.release_stream(self._comm_stream.cuda_stream)
            self._comm_stream = None

    def finish(self):
        """wait for the outstanding gradient all-reduces; returns the scale that turns SUM into MEAN.  Raises if a
        network that ran a training forward in this step never had its gradients reduced (ranks would diverge
        silently)."""
        arena = self.meta._arena
        for mid, module in self._expected.items():
            sl = self._slice(module, arena)
            if sl is not None and self._done.get(mid) != [(sl[0], sl[1])]:
                raise RuntimeError("data parallel: gradients of %s were not all-reduced in this step (forward without a "
                                   "completed backward?)" % type(module).__name__)
        if self._direct is not None:
            torch.cuda.current_stream(self._direct.device).wait_stream(self._comm_stream)
        for h in self.handles:
            h.wait()
        self.handles = []
        return 1.0 / self.world


def graph_dp_enabled():
    """hipGraph capture of data-parallel steps: on when every collective runs on the direct communicator and its
    start-up self-test (collectives on two streams inside a captured graph, replayed) passed; FSNET_AMD_GRAPH_DP=0
    keeps the step eager."""
    return os.environ.get("FSNET_AMD_GRAPH_DP", "1") != "0"
