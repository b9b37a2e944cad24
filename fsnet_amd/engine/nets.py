"""Manual forward/backward runners for the three networks on the hot path, written against the HIP
library only (conv igemm / wgrad, BN, pooling, upcat, heads).  The nn.Modules that own these
runners are pure parameter containers with the reference's state_dict names; autograd sees one
custom Function per network invocation.

Reference structure followed:
  ResNetRunner        vision_base/networks/models/backbone/resnet.py:21-50,53-89,199-213
  DepthDecoderRunner  monodepth/networks/models/heads/depth_encoder.py:45-66,119-139
  PoseDecoderRunner   monodepth/networks/models/heads/pose_decoder.py:26-45
"""
import os
import weakref

import torch

from ..hip import ops
from ..hip.binding import raw_stream
from ..hip.conv import ConvOp, run_specs
from .runtime import RT, grad_of

STAT_SLOTS = ops.STAT_SLOTS
# BatchNorm + ReLU between two convolutions of a residual block applied by the SECOND convolution while it stages its
# operand (and by its weight gradient), instead of a pass of its own: the normalised activation never reaches HBM.
# FSNET_AMD_BN_FOLD=0 keeps the pass (tests/test_bn_fold_gpu.py compares the two).
FOLD_BN = os.environ.get("FSNET_AMD_BN_FOLD", "1").strip().lower() not in ("0", "off", "false", "no")
# (The backward counterpart — the second pass of a BatchNorm's backward applied by the data gradient in front of it while it
# stages dY, FsConvArgs.pro_mode = 2 of rounds 3-4 — was built, pinned to the oracle and measured slower: two staged tensors,
# 172-198 registers, half the blocks per CU, +19-25 us for a 15 us pass.  Removed in round 5 with its kernel paths.)
# Always on since round 6 (their switches are gone; docs/LAB_r02_r05.md has the measurements): the BatchNorm-backward
# reduction fused into the epilogue of the data gradient that feeds it; the 1x1 / stride-2 downsample projection's data
# gradient inside the block's 3x3 / stride-2 data-gradient launch; a hand-off batch's bias gradients in one launch; the stem's
# BatchNorm + ReLU + max-pool as one pass with its backward's pooling gradient gathered inside the BatchNorm passes.


class StatsPool:
    """f64 scratch for BatchNorm batch statistics: one memset per network forward.  16 MB: the ResNet-50 pose encoder's
    forward takes 0.85 M doubles (two statistics groups x 8 slots x 2 x 26.5 k channels); a pool that runs out falls back
    to one zero-filled allocation per BatchNorm (45 fill launches per step at the 4 MB this started with)."""

    def __init__(self, device, capacity=1 << 21):
        self.buf = torch.zeros(capacity, dtype=torch.float64, device=device)
        self.off = 0
        # everything below `high` may hold sums: captured steps of different arrangements (the training hook's autotune
        # keeps two graphs for a while) each dirty their own extent whatever this object's bump pointer says
        self.high = 0
        self.dirty = False
        self.epoch = -1           # ops.PREZERO_EPOCH of the step-head zeroing that last covered this pool
        ops.register_prezero(self, StatsPool._prezero, device)

    def _prezero(self):
        """(ops.prezero_all, at the step's head) the used prefix is about to be zeroed with the rest of the step's scratch:
        the reset() at the start of the network's forward / backward then finds nothing to do"""
        self.off = 0
        self.epoch = ops.PREZERO_EPOCH[0]
        if not self.dirty:
            return []
        self.dirty = False
        return [self.buf[:self.high]]

    def reset(self):
        if self.dirty:
            self.buf[:self.high].zero_()     # everything beyond the high-water mark was never handed out: still zero
            self.dirty = False
        self.off = 0

    def span(self, a, b):
        """one contiguous view over two slices taken back to back (a first), or None"""
        if a is None or b is None or a.untyped_storage().data_ptr() != self.buf.untyped_storage().data_ptr() \
                or b.untyped_storage().data_ptr() != self.buf.untyped_storage().data_ptr():
            return None
        oa, ob = a.storage_offset(), b.storage_offset()
        if oa + a.numel() != ob:
            return None
        return self.buf[oa:ob + b.numel()]

    def take(self, C, groups=1):
        """[SLOTS][2][C] (one statistics group) or [G][SLOTS][2][C]"""
        n = groups * STAT_SLOTS * 2 * C
        shape = (STAT_SLOTS, 2, C) if groups == 1 else (groups, STAT_SLOTS, 2, C)
        if self.off + n > self.buf.numel():
            return torch.zeros(shape, dtype=torch.float64, device=self.buf.device)
        v = self.buf[self.off:self.off + n].view(shape)
        self.off += n
        self.high = max(self.high, self.off)
        self.dirty = True
        return v


_PENDING_JOIN = set()   # (chain stream, companion stream) pairs with weight-gradient work in flight
_PENDING_KEEP = []      # tensors the companion kernels still read: kept alive until the join (no record_stream
                        # bookkeeping in the allocator, and safe inside a hipGraph capture's private pool)
_DEFERRED = {}          # chain stream id -> (chain stream, [weight-gradient work items not yet handed over])
_CALLBACK_QUEUED = [None]     # id of the backward pass whose end-of-backward callback is queued
_ACTIVE_CHAINS = set()  # chain stream ids that deferred work during the running backward pass


_STREAM_OBJS = {}


def _current_stream(device=None):
    """torch's current Stream object, cached by raw handle (building one costs ~6 us; 60+ lookups per step)"""
    from ..hip.binding import raw_stream
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    key = (idx, raw_stream(idx))
    s = _STREAM_OBJS.get(key)
    if s is None:
        s = _STREAM_OBJS[key] = torch.cuda.current_stream(idx)
    return s


def _run_param_grads(*item, bias_later=None):
    """one layer's weight (+ bias) gradient — item = (op, dc, x, gw, gb, nb, pro) — or the same layer of two networks in
    one launch — item = ("lanes", [those tuples]) (fs_conv_wgrad2: one round of blocks, one slab arena, two dW)"""
    lanes = item[1] if item[0] == "lanes" else [item]
    run_specs([op.wgrad_spec(dc, x, gw, pro=pro) for (op, dc, x, gw, gb, nb, pro) in lanes])
    for (op, dc, x, gw, gb, nb, pro) in lanes:
        if gb is not None:
            if bias_later is not None:
                bias_later.append((dc, gb, nb))     # (the caller sums the batch's biases in one launch)
            else:
                ops.channel_sum(dc, gb, nb)


_INLINE_BIAS = {}       # chain stream id -> (chain stream, [(dc, gb, nb)]): bias sums of inline weight gradients not yet issued


def _inline_bias_list(dev):
    """data parallel: the weight gradients run inline on their chain — their bias sums are still collected and issued as
    one launch per gradient bucket (flush_inline_bias from DataParallelContext._reduce_range) instead of one per layer
    (18 launches of ~10 us on the decoders' chains: tools/probes/dp_world1.py)"""
    if not (RT.dp is not None and dev.type == "cuda"):
        return None
    cur = _current_stream(dev)
    return _INLINE_BIAS.setdefault(cur.cuda_stream, (cur, []))[1]


def drop_inline_bias():
    """a new step begins (DataParallelContext.begin_step): whatever a backward that raised left uncollected is not this
    step's gradient"""
    for ent in _INLINE_BIAS.values():
        ent[1].clear()


def flush_inline_bias(cur=None):
    """issue the collected bias sums of chain stream `cur` (default: every chain) on their chain"""
    for key in ([cur.cuda_stream] if cur is not None else list(_INLINE_BIAS.keys())):
        ent = _INLINE_BIAS.get(key)
        if ent is None or not ent[1]:
            continue
        chain, items = ent
        with torch.cuda.stream(chain):
            ops.channel_sum_multi(items)
        items.clear()


def accumulate_param_grads_multi(cls, ops_, dcs, xs, pros):
    """ConvLayer.accumulate_param_grads for the same layer of the lanes of an EncoderPass: the lanes whose weights are
    trained share one weight-gradient launch (deferred to the companion stream like a single layer's)"""
    lanes = []
    for cl, op, dc, x, pro in zip(cls, ops_, dcs, xs, pros):
        gw = grad_of(cl.m.weight)
        gb = grad_of(cl.m.bias) if cl.m.bias is not None else None
        if gw is None:                       # frozen layer: nothing to accumulate (a lone trainable bias: its sum)
            if gb is not None:
                ops.channel_sum(dc, gb, cl.m.bias.numel())
            continue
        lanes.append((op, dc, x, gw, gb, cl.m.bias.numel() if gb is not None else 0, pro))
    if not lanes:
        return
    item = lanes[0] if len(lanes) == 1 else ("lanes", lanes)
    dc0 = lanes[0][1]
    mode = RT.wgrad_streams if (dc0.is_cuda and RT.overlap and (RT.dp is None or RT.dp.wgrad_companions)) else 0
    if mode:
        _defer_param_grads(_current_stream(dc0.device), item)
        return
    _run_param_grads(*item, bias_later=_inline_bias_list(dc0.device))


def _drop_stale():
    """leftovers of a backward pass that raised (its end-of-backward callback never ran): not this pass's gradients"""
    if _CALLBACK_QUEUED[0] is not None and _CALLBACK_QUEUED[0] != torch._C._current_graph_task_id():
        for stale in _DEFERRED.values():
            stale[1].clear()
        _ACTIVE_CHAINS.clear()
        _LATE.clear()
        _AT_END.clear()


def _defer_param_grads(cur, item):
    _drop_stale()
    ent = _DEFERRED.get(cur.cuda_stream)
    if ent is None:
        ent = _DEFERRED[cur.cuda_stream] = (cur, [])
    ent[1].append(item)
    _ACTIVE_CHAINS.add(cur.cuda_stream)
    # runs once, when the autograd engine has executed every node of this backward pass and before it synchronises the
    # streams it used with the caller's stream.  (Keyed by the pass: a backward that died in an exception never ran its
    # callback, and must not keep the next one from queueing its own.)
    _queue_end_of_backward()
    if len(ent[1]) >= (RT.wgrad_flush_even if RT.even_chains else RT.wgrad_flush):
        flush_deferred(cur)


_BATCH_NO = [0]         # (numbering of the FSNET_AMD_MARKS=1 marks around each batch)
HANDOVERS = {"late": 0, "shared": 0}    # batches issued late / shared with the other chain's streams (tests read it)
_ENCODER_END = [False]  # the last two stages and the stem of an encoder's backward are being issued (EncoderPass.backward)
_AT_END = []            # [(event, chain, [(stream, items)])]: parts of a batch issued once every node of the pass has been
_LATE = []              # inside a hipGraph capture, put off: [(event on the chain, chain, companion, items, position)] batches,
                        # [(event, chain, None, fn, position)] calls (late_call)
_CHAINS_BEGUN = [None, []]    # backward pass (graph task id), chain streams whose backward has begun in it


def _issue_batch(chain, ws, mine, ev=None):
    """weight-gradient items of `chain` on stream `ws`, behind the chain (ev: behind an event recorded on it earlier)"""
    if ev is None:
        ws.wait_stream(chain)                   # one cross-stream edge per batch
    else:
        ws.wait_event(ev)
    with torch.cuda.stream(ws):
        if RT._marks is not None:
            _BATCH_NO[0] += 1
            tag = "wg.%s.%d" % ("pose" if RT.is_side(chain) else "depth", _BATCH_NO[0])
            RT.mark(tag + ".start")
        sums = []
        for it in mine:
            _run_param_grads(*it, bias_later=sums)
        if sums:
            ops.channel_sum_multi(sums)         # the batch's bias gradients in one launch
        if RT._marks is not None:
            RT.mark(tag + ".end(%d)" % len(mine))
    _PENDING_JOIN.add((chain, ws))


def _position(chain):
    """token of `chain`'s position in the running capture (fs_capture_position: changes whenever the stream captures a node)"""
    import ctypes as C
    from ..hip.binding import check, lib
    tok = C.c_ulonglong(0)
    check(lib.fs_capture_position(C.c_void_p(chain.cuda_stream), C.byref(tok)), "fs_capture_position")
    return tok.value


def issue_late(chain=None, advanced=False):
    """issue what was put off on `chain` (default: every chain) inside a capture, in the order it was put off: weight-
    gradient batches (flush_deferred) and gradient-bucket reductions (late_call, from DataParallelContext._reduce_range).
    advanced: only what was put off at an EARLIER position of the chain than its current one — the chain has captured its
    next kernel since, which is all the putting-off is for (flush_deferred); what was put off where the chain still stands
    stays."""
    if not _LATE:
        return
    mine = [e for e in _LATE if chain is None or e[1].cuda_stream == chain.cuda_stream]
    if advanced and mine:
        here = _position(chain)
        mine = [e for e in mine if e[4] != here]
    if not mine:
        return
    ids = {id(e) for e in mine}
    _LATE[:] = [e for e in _LATE if id(e) not in ids]
    for ent in mine:
        if ent[2] is None:
            ent[3](ent[0])                      # fn(event recorded on the chain where the call was put off)
        else:
            _issue_batch(ent[1], ent[2], ent[3], ev=ent[0])


def issue_advanced(device):
    """a cheap place to let go of what the current chain put off (block and level boundaries of the backward passes): under
    data parallelism a bucket's all-reduce takes its place in the communicator's launch order when it is ISSUED, and one
    issued at the end of the pass would run after every SyncBN exchange of the backward instead of beside it"""
    if _LATE:
        _drop_stale()
        issue_late(_current_stream(device), advanced=True)


def late_call(chain, fn):
    """inside a capture: call fn(event) when every node of the backward pass has been issued (see flush_deferred) —
    `event` is recorded on `chain` now, and what fn issues on another stream behind it becomes a LATER successor of the
    chain's last node than the chain's next kernel.  Returns False, and does nothing, outside a capture or an autograd
    pass."""
    if not (RT.wgrad_late and torch.cuda.is_current_stream_capturing() and torch._C._current_graph_task_id() >= 0):
        return False
    _drop_stale()
    issue_late(chain, advanced=True)
    ev = torch.cuda.Event()
    ev.record(chain)
    _LATE.append((ev, chain, None, fn, _position(chain)))
    _queue_end_of_backward()
    return True


def flush_deferred(cur=None, now=False):
    """hand the collected weight-gradient work of chain stream `cur` (default: all chains) to its companion.

    Inside a hipGraph capture the batch's kernels are ISSUED later (`now`: at once) — at the chain's next hand-over or block
    boundary that finds it further on (issue_late(advanced=True)), at the latest when the whole backward pass has been
    issued — behind the event recorded here.  Dependencies are the same; what changes is the ORDER of the edges that leave
    the chain's last node: the chain's next kernel becomes its first successor, the batch its second.  The HIP graph executor
    (ROCm 7.2) hands out its 4 streams by a depth-first walk in which a node's first successor stays on the node's stream and
    the k-th further one goes k streams on (mod 4); each stream runs its nodes in the order they were captured.  With the
    batch first, the weight gradients inherited the chain's stream and the chain hopped to one where it queued behind
    whatever the other chain had there (docs/LAB_r06.md, "the executor's stream assignment": DEBUG_HIP_GRAPH_DOT_PRINT
    dumps read with tools/probes/graph_streams.py)."""
    late = (RT.wgrad_late and not now and torch.cuda.is_current_stream_capturing()
            and torch._C._current_graph_task_id() >= 0)
    for key in ([cur.cuda_stream] if cur is not None else list(_DEFERRED.keys())):
        ent = _DEFERRED.get(key)
        if ent is None or not ent[1]:
            continue
        chain, items = ent
        issue_late(chain, advanced=late)        # (what was put off goes first: the companion runs batches in this order)
        ws = RT.companion_stream(chain.device, chain)[1]
        ev = None
        capturing = late or torch.cuda.is_current_stream_capturing()
        if capturing:
            ev = torch.cuda.Event()
            ev.record(chain)
        mine = list(items)
        if (RT.wgrad_balance and RT.even_chains and capturing and _ENCODER_END[0] and RT.dp is None and not RT.is_side(chain)
                and len(items) >= 2 and torch._C._current_graph_task_id() >= 0):
            # the end of the main chain's encoder: the chain's companion still has the depth decoder's weight gradients to
            # run when the encoder's arrive, and ends the step with them after the pose chain and its companion have
            # finished.  Half of each batch from layer2 on goes to the pose chain's stream, issued at the end of the pass
            # like everything else — behind the pose chain's own work; as the third successor of the hand-over point the
            # share lands on the executor's stream of the pose chain.  (Shares of the encoder's earlier batches run
            # beside the depth chain rather than after it: measured, they slow it more than they relieve the companion.)
            side = RT.side_stream(chain.device)
            others = [side, RT.companion_stream(chain.device, side)[1]][:RT.wgrad_balance]
            n = 1 + len(others)
            _AT_END.append((ev, chain, [(o, items[1 + k::n]) for k, o in enumerate(others)]))
            HANDOVERS["shared"] += 1
            mine = items[0::n]
        if late:
            _LATE.append((ev, chain, ws, mine, _position(chain)))
            HANDOVERS["late"] += 1
        else:
            _issue_batch(chain, ws, mine, ev=ev)
        _PENDING_KEEP.extend(items)
        ent[1].clear()
    if not late:
        issue_late(cur)


def chain_ends(device):
    """a chain has issued its last kernel (the end of an encoder's backward).  Inside a capture, with hand-overs put off: one
    empty launch on the chain, so that the last hand-over point has a first successor that stays on the chain's stream like
    every other one — otherwise the last batch inherits the chain's executor stream and the share that follows it the
    companion's, behind the companion's backlog."""
    if (RT.wgrad_late and (_LATE or _AT_END) and device.type == "cuda" and torch.cuda.is_current_stream_capturing()
            and torch._C._current_graph_task_id() >= 0):
        RT.nop(device)
        issue_late(_current_stream(device))


def chain_begins(device):
    """called where a network's backward begins on its chain (the decoders' autograd nodes).  Inside a hipGraph capture,
    when the SECOND chain of the pass begins, one empty launch is put on the first chain's companion, behind the second
    chain's pending dependencies — a third successor of the loss backward's last node, between the two chains' first
    kernels.  By the executor's rule (flush_deferred) the first chain then keeps that node's stream s, its weight gradients
    get s+1, the second chain s+2 and its weight gradients s+3: four streams, one each.  Without it the second chain
    shares s+1 with the first chain's weight gradients, behind them in launch order."""
    if not (RT.wgrad_late and device.type == "cuda" and RT.overlap and RT.wgrad_streams
            and torch.cuda.is_current_stream_capturing()):
        return
    task = torch._C._current_graph_task_id()
    if task < 0:
        return
    _drop_stale()
    if _CHAINS_BEGUN[0] != task:
        _CHAINS_BEGUN[0], _CHAINS_BEGUN[1] = task, []
    cur = _current_stream(device)
    begun = _CHAINS_BEGUN[1]
    if any(c.cuda_stream == cur.cuda_stream for c in begun):
        return
    begun.append(cur)
    if len(begun) == 2:
        first = begun[0]
        ws = RT.companion_stream(device, first)[1]
        ws.wait_stream(cur)
        with torch.cuda.stream(ws):
            RT.nop(device)
        _PENDING_JOIN.add((first, ws))


def _end_of_backward():
    _CALLBACK_QUEUED[0] = None
    flush_tail()
    flush_inline_bias()
    flush_deferred(now=True)
    for ev, chain, parts in _AT_END:
        for ws, mine in parts:
            if mine:
                _issue_batch(chain, ws, mine, ev=ev)
    _AT_END.clear()
    _ENCODER_END[0] = False
    _ACTIVE_CHAINS.clear()
    _BATCH_NO[0] = 0
    join_companions()


def join_companions():
    """every chain stream waits for its companion.  Inside a hipGraph capture the join is left to
    join_companions_final(): joining a companion into a stream that is itself a fork of the capture stream and
    then joining that one crashes hipStreamEndCapture on ROCm 7.2 (tools/probes/graph_fork_probe.py, variants
    C/D), joining every companion straight into the capture stream does not (variant G)."""
    if not _PENDING_JOIN:
        return
    if torch.cuda.is_current_stream_capturing():
        return
    for cur, ws in list(_PENDING_JOIN):
        cur.wait_stream(ws)
    _PENDING_JOIN.clear()
    _PENDING_KEEP.clear()


_TAIL_SINK = [None]     # while a network's backward collects its weight gradients for the tail: the list they go to
_TAIL = []              # [(event on the network's chain, device, items, runner)] waiting for the end of the backward pass


def _queue_end_of_backward():
    task = torch._C._current_graph_task_id()
    if _CALLBACK_QUEUED[0] != task:
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
        _CALLBACK_QUEUED[0] = task


def defer_to_tail(items, runner, device):
    """data parallel, weight gradients "tail" (DataParallelContext.wgrad_mode): the depth decoder's weight gradients — its
    chain goes on with the depth encoder's backward and ends the step, the pose chain's stream finishes earlier — are
    issued on the pose chain's stream once every node of the backward pass has been issued (flush_tail), behind that
    chain's own work; the decoder's gradient bucket follows them there."""
    if _TAIL and _CALLBACK_QUEUED[0] != torch._C._current_graph_task_id():
        _TAIL.clear()                        # leftovers of a backward pass that raised
    chain = _current_stream(device)
    ev = torch.cuda.Event()
    ev.record(chain)
    runner.tail_pending = True
    _TAIL.append((ev, chain, items, runner))
    _queue_end_of_backward()


def flush_tail():
    for ev, chain, items, runner in _TAIL:
        side = RT.side_stream(chain.device)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            sums = []
            for it in items:
                _run_param_grads(*it, bias_later=sums)
            if sums:
                ops.channel_sum_multi(sums)
            runner.tail_pending = False
            if RT.dp is not None and runner.m._pending == 0:
                RT.dp.grads_ready(runner.m)          # the bucket's all-reduce picks up from this stream
        _PENDING_JOIN.add((chain, side))             # (joined, and the operands released, like a companion's batch)
        _PENDING_KEEP.extend(items)
    _TAIL.clear()


def pending_companions():
    """companion streams with weight-gradient work in flight in this step"""
    return {ws for _, ws in _PENDING_JOIN}


def join_companions_final():
    """the current stream waits for every companion with work in flight (before the optimizer reads gradients)"""
    flush_inline_bias()
    flush_deferred(now=True)
    issue_late()
    if not _PENDING_JOIN:
        return
    cur = torch.cuda.current_stream()
    for ws in {ws for _, ws in _PENDING_JOIN}:
        cur.wait_stream(ws)
    _PENDING_JOIN.clear()
    _PENDING_KEEP.clear()


_PACK_REGISTRY = {}     # (dtype, device) -> [weakref to ConvLayer]: a dropped model leaves no work behind
_PACK_TABLES = {}       # (dtype, device, tuple of pointers) -> (device table, n, total_blocks)


def pack_everything(arena=None):
    """all registered (dtype, device) groups — called on the main stream before work is forked to a side stream.
    With an arena: only the convolutions whose master weights live in it (one model's step never touches, or
    captures pointers of, another model's layers)."""
    for key in list(_PACK_REGISTRY.keys()):
        pack_all(key, arena)


_PACK_PENDING = {}      # device index -> (pack stream, handles of the streams that already wait for it)


def pack_everything_async(arena):
    """pack_everything() on a stream of its own, forked from the current one: the step's first ~0.1 ms — staging the
    batch, zeroing the gradient arena, converting the images to NHWC, the image-only loss inputs — needs no weights, and
    the 70 us re-pack of every convolution's MFMA operands after the optimizer step ran in front of all of it.  Every
    stream that launches a convolution waits for the pack stream once (ConvLayer.ready -> join_pack)."""
    dev = arena.data.device if arena is not None else None
    if not (RT.pack_overlap and RT.overlap and dev is not None and dev.type == "cuda"):
        _PACK_PENDING.clear()
        pack_everything(arena)
        if arena is not None:
            arena.flush_zero()            # (a gradient memset the training hook left to the scratch-zeroing launch)
        return
    cur = _current_stream(dev)
    ps = RT.pack_stream(dev)
    ps.wait_stream(cur)
    with torch.cuda.stream(ps):
        pack_everything(arena)
        ops.prezero_all(dev)          # ... and the step's scratch, one launch instead of a fill here and there along the chains
    arena.flush_zero()                # (nothing to do: prezero_all took the gradient arena's memset)
    _PACK_PENDING[dev.index] = (ps, {ps.cuda_stream})


def join_pack(device):
    """the current stream waits (once per step) for the step's weight re-pack"""
    ent = _PACK_PENDING.get(device.index)
    if ent is None:
        return
    cur = _current_stream(device)
    if cur.cuda_stream in ent[1]:
        return
    cur.wait_stream(ent[0])
    ent[1].add(cur.cuda_stream)


def pack_all(key, arena=None):
    """Re-pack the MFMA weight operands of every registered conv whose master weights changed, in ONE launch
    (fs_pack_weights_multi).  The descriptor table is cached while the pointers stay the same."""
    import ctypes as C
    from ..hip.binding import FsPackDesc, lib, check, stream_ptr
    from ..hip.conv import dtype_code
    dtype, device = key
    refs = _PACK_REGISTRY.get(key, [])
    alive = [(r, r()) for r in refs]
    if any(l is None for _, l in alive):
        refs[:] = [r for r, l in alive if l is not None]
    layers = [l for _, l in alive if l is not None and l._version() != l._packed and l.m.weight.is_cuda
              and (arena is None or arena.owns(l.m.weight))]
    if not layers:
        return
    # everything a descriptor encodes: a freed model's successor can land on the same master-weight and forward-
    # operand addresses with its dgrad operand elsewhere (seen as a rare wrong feature gradient in a long test run:
    # the stale table packed into the old model's freed dgrad buffers and left the new ones unpacked)
    sig = tuple((l.m.weight.data_ptr(), l._op.w_f.data_ptr(), l._op.w_d.data_ptr() if l._op.need_dgrad else 0,
                 l._op.Co, l._op.Ci, l._op.R, l._op.S, l._op.stride) for l in layers)
    ent = _PACK_TABLES.get((key, sig))
    if ent is None:
        arr = (FsPackDesc * len(layers))()
        blocks = 0
        for d, l in zip(arr, layers):
            op, w = l._op, l.m.weight
            assert w.dtype == torch.float32 and w.data.is_contiguous()
            d.w, d.dst_f = w.data_ptr(), op.w_f.data_ptr()
            d.Co, d.Ci, d.R, d.S = op.Co, op.Ci, op.R, op.S
            d.rows_f, d.cs_f, d.k_f = op.Co_p, op.Ci_p, op.kf_p
            if op.need_dgrad:
                d.dst_d, d.rows_d, d.cs_d, d.k_d = op.w_d.data_ptr(), op.rows_d, op.Co_p, op.kd_p
            else:
                d.dst_d, d.rows_d, d.cs_d, d.k_d = None, 0, 1, 1
            d.tap_order_d = 1 if getattr(op, "s2_classes", False) else 0
            d.block_start = blocks
            blocks += int(lib.fs_pack_tile_blocks(op.Co, op.Ci, op.R, op.S))
        raw = bytes(arr)
        tab = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        # (never evicted: a captured hipGraph may hold the table's address; a table is ~100 B per conv)
        ent = _PACK_TABLES[(key, sig)] = (tab, len(layers), blocks)
    tab, n, blocks = ent
    check(lib.fs_pack_weights_multi(tab.data_ptr(), n, blocks, dtype_code(dtype), stream_ptr()), "pack_weights_multi")
    for l in layers:
        l._after_pack(device)


class ConvLayer:
    """Binds an nn.Conv2d parameter container to its device plan (ConvOp)."""

    def __init__(self, conv, valid_over_padded=False, need_dgrad=True):
        self.m = conv
        k = conv.kernel_size
        self.R, self.S = int(k[0]), int(k[1])
        s = conv.stride
        self.stride = int(s[0]) if isinstance(s, (tuple, list)) else int(s)
        p = conv.padding
        pad = int(p[0]) if isinstance(p, (tuple, list)) else int(p)
        # 'replicate' convs run as a valid conv over a buffer the producer already padded
        self.pad = 0 if valid_over_padded else pad
        self.need_dgrad = need_dgrad
        self._op = None
        self._key = None
        self._owner = None
        self._packed = None
        self._bias = None

    def ready(self, dtype, device):
        key = (dtype, device)
        if _PACK_PENDING:
            join_pack(device)
        if self._key != key or self._owner != id(self):
            # (_owner: a copy.deepcopy of the module copies this object's state but is not in the pack registry —
            # without the check the copy's MFMA operands would never follow its own weights)
            w = self.m.weight
            self._op = ConvOp(w.shape[1], w.shape[0], self.R, self.S, self.stride, self.pad, dtype, device,
                              need_dgrad=self.need_dgrad)
            self._key, self._packed, self._owner = key, None, id(self)
            _PACK_REGISTRY.setdefault(key, []).append(weakref.ref(self))
        if self._version() != self._packed:
            pack_all(key)          # one launch for every stale conv of the model
        return self._op

    def _version(self):
        w = self.m.weight
        return (w._version, RT.weights_epoch, w.data_ptr())

    def _after_pack(self, device):
        b = self.m.bias
        if b is not None:
            if self._op.Co_p == b.numel():
                self._bias = b.data
            else:
                if self._bias is None or self._bias.numel() != self._op.Co_p or self._bias.data_ptr() == b.data_ptr():
                    self._bias = torch.zeros(self._op.Co_p, dtype=torch.float32, device=device)
                self._bias[: b.numel()].copy_(b.data)
        self._packed = self._version()

    @property
    def bias(self):
        return self._bias

    def accumulate_param_grads(self, op, dc, x, pro=None):
        """wgrad + bias grad into the parameters' gradient buffers.  They feed nothing downstream in the
        backward pass, so they run on a companion stream while the chain continues with the dgrad.
        pro: x is a raw convolution output whose BatchNorm (+ ReLU) the weight gradient applies itself."""
        gw = grad_of(self.m.weight)
        gb = grad_of(self.m.bias) if self.m.bias is not None else None
        if gw is None:                       # frozen layer: nothing to accumulate (a lone trainable bias: its sum)
            if gb is not None:
                ops.channel_sum(dc, gb, self.m.bias.numel())
            return
        item = (op, dc, x, gw, gb, self.m.bias.numel() if gb is not None else 0, pro)
        if _TAIL_SINK[0] is not None:
            _TAIL_SINK[0].append(item)       # (data parallel, "tail": issued when the backward pass has been issued)
            return
        # (data parallel: inline on the chain stream, so that one event after a stage's last weight gradient covers
        # the arena slice its gradient bucket reduces — also inside a captured step)
        mode = RT.wgrad_streams if (dc.is_cuda and RT.overlap and (RT.dp is None or RT.dp.wgrad_companions)) else 0
        if mode:
            _defer_param_grads(_current_stream(dc.device), item)
            return
        _run_param_grads(*item, bias_later=_inline_bias_list(dc.device))


def bn_tensors(bn):
    """the module's tensors as a dict, cached on the module while its storage stays where it is (five
    nn.Module.__getattr__ lookups per BatchNorm call add up on the host side)"""
    c = bn.__dict__.get("_fs_tensors")
    w = bn.weight
    if c is None or c[0] != w.data_ptr() or c[1] != bn.running_mean.data_ptr():
        d = {"weight": w.data, "bias": bn.bias.data, "running_mean": bn.running_mean,
             "running_var": bn.running_var, "num_batches_tracked": bn.num_batches_tracked}
        c = (w.data_ptr(), bn.running_mean.data_ptr(), d)
        bn.__dict__["_fs_tensors"] = c
    return c[2]


def _dp_stats(stats):
    if RT.dp is not None:
        RT.dp.allreduce_small(stats)
        return RT.dp.world
    return 1


_BWD_POOLS = {}


def bwd_pool_reset(device):
    """the current stream's pool of BatchNorm-backward sums, zeroed (one memset per network backward outside a training step;
    inside one: by the step's one zeroing launch)"""
    key = (device, raw_stream(device.index))
    p = _BWD_POOLS.get(key)
    if p is None:
        p = _BWD_POOLS[key] = StatsPool(device)
    if p.epoch != ops.PREZERO_EPOCH[0]:
        p.reset()
    # (else: zeroed at the head of the running step with the rest of its scratch, and a network earlier on this chain has
    # taken from it since — the depth decoder before the depth encoder: the next network's sums follow on, no fill launch)
    return p


def _bwd_sums(c, st):
    """zeroed f64 [groups][SLOTS][2][C] from the current stream's backward pool"""
    return _BWD_POOLS[(c.device, raw_stream(c.device.index))].take(c.shape[-1], st.groups)


def _bn_bwd(dout, y, c, bn, st, H, W, relu=True, fold=False, g_out=None, sums=None):
    """sums given: dout is already ReLU-masked and the sums are accumulated (fused into the producing dgrad)"""
    dc = torch.empty_like(c)
    reduced = sums is not None
    if sums is None:
        sums = _bwd_sums(c, st)
    # (eval-mode BatchNorm: st.count is inf — no batch-statistics terms in dx, hence no exchange of the sums either)
    sync = RT.dp is not None and st.count != float("inf")
    ops.bn_backward(dout, y, c, bn.weight.data, st, dc, grad_of(bn.weight), grad_of(bn.bias), H, W, relu=relu,
                    fold=fold, g_out=g_out, sums=sums, sums_zeroed=True, reduced=reduced,
                    allreduce=(RT.dp.allreduce_small if sync else None))
    return dc


def _check_train_bn(bn, what):
    if not bn.training:
        raise NotImplementedError(
            "%s: BatchNorm in eval mode inside a training step is only implemented for the ResNet encoders "
            "(norm_eval / frozen_stages); the reference has no such switch for this module" % what)


# ==============================================================================================
# ResNet encoder
# ==============================================================================================
def _exchange(pool, tensors, out_of_place=False):
    """Data-parallel exchange (SUM over the ranks) of f64 statistics slices taken from `pool`: slices that lie back to
    back in the pool's buffer — the downsample branch's and the main branch's, the depth encoder's and the pose encoder's
    of the same layer — travel in ONE collective.  In place, or (out_of_place) into fresh tensors that are returned in
    the order of `tensors` while the inputs keep the local sums.  Returns (world size, results)."""
    if RT.dp is None:
        return 1, list(tensors)
    base = pool.buf.untyped_storage().data_ptr() if pool is not None else None
    idx = [i for i, t in enumerate(tensors) if t is not None]
    inside = [i for i in idx if base is not None and tensors[i].untyped_storage().data_ptr() == base]
    res = list(tensors)
    runs, cur = [], []
    for i in sorted(inside, key=lambda i: tensors[i].storage_offset()):
        if cur and tensors[cur[-1]].storage_offset() + tensors[cur[-1]].numel() == tensors[i].storage_offset():
            cur.append(i)
        else:
            if cur:
                runs.append(cur)
            cur = [i]
    if cur:
        runs.append(cur)
    for run in runs:
        lo = tensors[run[0]].storage_offset()
        hi = tensors[run[-1]].storage_offset() + tensors[run[-1]].numel()
        view = pool.buf[lo:hi]
        if out_of_place:
            out = torch.empty_like(view)
            RT.dp.allreduce_small(view, out=out)
            for i in run:
                o = tensors[i].storage_offset() - lo
                res[i] = out[o:o + tensors[i].numel()].view(tensors[i].shape)
        else:
            RT.dp.allreduce_small(view)
    for i in idx:
        if i in inside:
            continue
        if out_of_place:
            res[i] = torch.empty_like(tensors[i])
            RT.dp.allreduce_small(tensors[i], out=res[i])
        else:
            RT.dp.allreduce_small(tensors[i])
    return RT.dp.world, res


class ResNetRunner:
    """One encoder's layers bound to their device plans.  Execution is EncoderPass's: forward / backward here run a
    one-lane pass; MonoDepthMeta runs the depth and the pose encoder as the two lanes of one pass."""

    def __init__(self, module):
        self.m = module
        self.stem = ConvLayer(module.conv1, need_dgrad=False)
        self.stages = []
        for i in range(module.num_stages):
            blocks = []
            for blk in getattr(module, "layer%d" % (i + 1)):
                if hasattr(blk, "conv3"):
                    units = [(ConvLayer(blk.conv1), blk.bn1), (ConvLayer(blk.conv2), blk.bn2),
                             (ConvLayer(blk.conv3), blk.bn3)]
                else:
                    units = [(ConvLayer(blk.conv1), blk.bn1), (ConvLayer(blk.conv2), blk.bn2)]
                ds = None
                if blk.downsample is not None:
                    ds = (ConvLayer(blk.downsample[0]), blk.downsample[1])
                blocks.append((units, ds))
            self.stages.append(blocks)
        self.pool = None
        self._solo = None

    def signature(self, train):
        """what two encoders must share to run as the lanes of one pass: layer shapes, BatchNorm modes, which parameters
        are trained (the pass takes every control-flow decision once for both lanes)"""
        sig = []

        def unit(cl, bn):
            w = cl.m.weight
            sig.append((tuple(w.shape[2:]), w.shape[0], cl.stride, cl.pad, bool(train and bn.training), w.requires_grad,
                        bn.weight.requires_grad, bn.bias.requires_grad))
        unit(self.stem, self.m.bn1)
        sig[0] = sig[0][:1] + ("stem",) + sig[0][2:]
        for blocks in self.stages:
            for units, ds in blocks:
                for cl, bn in units:
                    unit(cl, bn)
                    sig[-1] = sig[-1] + (cl.m.weight.shape[1],)
                sig.append(None if ds is None else "ds")
                if ds is not None:
                    unit(*ds)
        return tuple(sig)

    def forward(self, x, train, groups=1):
        """x: NHWC [N,H,W,Ci_p] in the compute dtype.  Returns (features NHWC x5, ctx).
        groups G > 1 (training): x stacks G independent calls of the module along N — BatchNorm statistics,
        running-statistic updates and gradients are those of G separate calls in that order, the launches are
        shared (the pose encoder's two image pairs run as one pass)."""
        if self._solo is None:
            self._solo = EncoderPass([self])
        feats, ctx = self._solo.forward([x], train, [groups])
        return feats[0], ctx

    def backward(self, ctx, gfeats):
        """gfeats: list of 5 NHWC dense gradients (or None).  Accumulates parameter gradients."""
        self._solo.backward(ctx, [gfeats])


class EncoderPass:
    """Forward / backward of ONE ResNet — or of TWO of the same architecture in lockstep ("lanes"): the depth encoder on
    the target images and the pose encoder on the stacked image pairs (monodepth2_model.py:24-43) run the same layer
    shapes after their stems (resnet.py:199-213), so every launch of the pass carries both lanes' problems through the
    fs_*2 entry points (include/fsnet_hip.h): one launch's fixed cost and, under data parallelism, one SyncBN exchange
    per layer for both networks.  Everything per lane — tensors, weights, BatchNorm modules, statistics groups — is a
    list over the lanes; every control-flow decision is taken once (ResNetRunner.signature guarantees they agree).
    Numerically each lane is exactly its own pass: no arithmetic depends on the pairing."""

    def __init__(self, runners, need_feat0=None):
        self.R = list(runners)
        self.nl = len(self.R)
        self.pool = None
        # per lane: is features[0] (the stem's activation, resnet.py:201-204) read by anyone?  The pose decoder only takes
        # the last feature (pose_decoder.py:26-37): the fused stem pass then never stores that lane's 96 x 320 activation
        self.need_feat0 = list(need_feat0) if need_feat0 is not None else [True] * self.nl

    # ------------------------------------------------------------------ helpers
    def _lanes(self, per_runner):
        """[f(runner) for each lane]"""
        return [per_runner(r) for r in self.R]

    def _ready(self, cls, xs):
        return [cl.ready(x.dtype, x.device) for cl, x in zip(cls, xs)]

    # ------------------------------------------------------------------ forward
    def _unit_fwd_fold(self, cls, bns, xs, pros):
        """convolution + the batch statistics of its BatchNorm only (training mode): returns (raw outputs, BnStates, what
        the consumer needs to finalise them) — the consuming convolution derives scale / shift from the sums in its own
        prologue, applies scale * c + shift and the ReLU while staging, and its block 0 fills the BnState (mean, invstd,
        scale, shift: read by the backward) and updates the running statistics: no launch between the two convolutions"""
        ops_ = self._ready(cls, xs)
        stats = [self.pool.take(op.Co_p, G) for op, G in zip(ops_, self.groups)]
        specs = [op.forward_spec(x, stats=s, stat_groups=G, pro=p) for op, x, s, G, p in zip(ops_, xs, stats, self.groups, pros)]
        run_specs(specs)
        world, _ = _exchange(self.pool, stats)
        cs, sts, fins = [], [], []
        for op, x, sp, s, G, bn in zip(ops_, xs, specs, stats, self.groups, bns):
            N, H, W, _ = x.shape
            Ho, Wo = op.out_hw(H, W)
            st = ops.BnState(op.Co_p, x.device, G, affine=True)
            st.count = float((N // G) * Ho * Wo * world)
            cs.append(sp.out); sts.append(st); fins.append((s, bn_tensors(bn), st.count, True))
        return cs, sts, fins

    def _can_fold(self, bns, nxt_cls, c_shapes, xs, train):
        """unit -> next unit of a block: may the BatchNorm + ReLU in between be folded into the next convolution?"""
        if not (FOLD_BN and train and bns[0].training):
            return False
        for bn, ncl, (N, H, W), x, G in zip(bns, nxt_cls, c_shapes, xs, self.groups):
            nop = ncl.ready(x.dtype, x.device)
            if not (nop.can_fold_input(N, H, W) and nop.can_fuse_bn_bwd(N, H, W, G) and N * H * W >= 1):
                return False
        return True

    def _unit_fwd(self, cls, bns, xs, train, relu=True, res=None, ds_c=None, ds_stats=None, ds_bn=None, pros=None):
        nl = self.nl
        ops_ = self._ready(cls, xs)
        # a BatchNorm in eval mode inside a training step (norm_eval / frozen stages, resnet.py:169-197): running
        # statistics, no update of them, one statistics group; its backward is the batch-statistics formula with the
        # mean terms switched off (count = inf), dgamma / dbeta from the same sums
        bt = train and bns[0].training
        assert all((train and bn.training) == bt for bn in bns), "lanes disagree on a BatchNorm's mode"
        assert ds_bn is None or all((train and b.training) == bt for b in ds_bn), "main and downsample BatchNorm modes differ"
        Gs = [G if bt else 1 for G in self.groups]
        stats = [self.pool.take(op.Co_p, G) if bt else None for op, G in zip(ops_, Gs)]
        pros = pros if pros is not None else [None] * nl
        specs = [op.forward_spec(x, stats=s, stat_groups=G, pro=p) for op, x, s, G, p in zip(ops_, xs, stats, Gs, pros)]
        run_specs(specs)
        cs = [sp.out for sp in specs]
        # block end with a downsample branch: its statistics were taken right before this conv's and are consumed by the
        # same bn_apply — one exchange for both (and for both lanes)
        world = 1
        if bt:
            world, _ = _exchange(self.pool, (list(ds_stats) if ds_stats is not None else []) + stats)
        ys, sts, st2s, bspecs = [], [], [], []
        for l in range(nl):
            op, x, c = ops_[l], xs[l], cs[l]
            N, H, W, _ = x.shape
            Ho, Wo = op.out_hw(H, W)
            y = torch.empty(N, Ho, Wo, op.Co_p, dtype=x.dtype, device=x.device)
            st = ops.BnState(op.Co_p, x.device, Gs[l])
            st2 = ops.BnState(op.Co_p, x.device, Gs[l]) if ds_bn is not None else None
            count = (N // Gs[l]) * Ho * Wo * world if (bt or not train) else float("inf")
            bspecs.append(ops.bn_apply_spec(
                c, stats[l], bn_tensors(bns[l]), st, y, Ho, Wo, count, relu=relu,
                res=(ds_c[l] if ds_bn is not None else (res[l] if res is not None else None)),
                stats2=(ds_stats[l] if (bt and ds_stats is not None) else None),
                bn2=(bn_tensors(ds_bn[l]) if ds_bn is not None else None), st2=st2, track=bt, groups=Gs[l]))
            ys.append(y); sts.append(st); st2s.append(st2)
        run_specs(bspecs)
        return cs, ys, sts, st2s

    def _stem_fusable(self, cls, bns, xs, train):
        """stem BatchNorm + ReLU + max-pool as one pass (fs_bn_apply with FsBnApplyArgs.pool_y) and, backwards, the pooling
        gradient gathered inside both BatchNorm-backward passes (FsBnBwdArgs.pool_dy): training-mode BatchNorm and an
        even convolution output"""
        if not (train and all(bn.training for bn in bns)):
            return False
        for cl, x in zip(cls, xs):
            Ho, Wo = cl.ready(x.dtype, x.device).out_hw(x.shape[1], x.shape[2])
            if Ho % 2 or Wo % 2 or Ho < 2 or Wo < 2:
                return False
        return True

    def _stem_fwd_fused(self, cls, bns, xs):
        """-> (raw stem outputs, activations (None where no one reads features[0]), BnStates, [(pooled, argmax codes)])"""
        ops_ = self._ready(cls, xs)
        stats = [self.pool.take(op.Co_p, G) for op, G in zip(ops_, self.groups)]
        specs = [op.forward_spec(x, stats=s, stat_groups=G) for op, x, s, G in zip(ops_, xs, stats, self.groups)]
        run_specs(specs)
        cs = [sp.out for sp in specs]
        world, _ = _exchange(self.pool, stats)
        ys, sts, pooled, bspecs = [], [], [], []
        for l in range(self.nl):
            c, G = cs[l], self.groups[l]
            N, H, W, C = c.shape
            y = torch.empty_like(c) if self.need_feat0[l] else None
            pl = (torch.empty(N, H // 2, W // 2, C, dtype=c.dtype, device=c.device),
                  torch.empty(N, H // 2, W // 2, C, dtype=torch.uint8, device=c.device))
            st = ops.BnState(C, c.device, G)
            bspecs.append(ops.bn_apply_spec(c, stats[l], bn_tensors(bns[l]), st, y, H, W, (N // G) * H * W * world, relu=True,
                                            track=True, groups=G, pool=pl))
            ys.append(y); sts.append(st); pooled.append(pl)
        run_specs(bspecs)
        return cs, ys, sts, pooled

    def forward(self, xs, train, groups):
        """xs: per lane NHWC [N,H,W,Ci_p] in the compute dtype; groups: per lane statistics groups (see
        ResNetRunner.forward).  Returns (per lane: 5 features NHWC, ctx)."""
        nl = self.nl
        dev = xs[0].device
        if self.pool is None or self.pool.buf.device != dev:
            # (a two-lane pass object lives for one step: its pool lives with the first lane's runner)
            keep = self.R[0].__dict__.get("_pass_pool")
            if keep is None or keep.buf.device != dev:
                keep = self.R[0]._pass_pool = StatsPool(dev)
            self.pool = keep
        self.pool.reset()
        self.groups = [g if train else 1 for g in groups]
        assert all(x.shape[0] % g == 0 for x, g in zip(xs, self.groups))
        ctx = {"x": xs, "blocks": [], "train": train}
        stem_cls, stem_bns = self._lanes(lambda r: r.stem), self._lanes(lambda r: r.m.bn1)
        if self._stem_fusable(stem_cls, stem_bns, xs, train):
            c0, y0, st0, pooled = self._stem_fwd_fused(stem_cls, stem_bns, xs)
            ctx["stem_fused"] = True
        else:
            c0, y0, st0, _ = self._unit_fwd(stem_cls, stem_bns, xs, train)
            pooled = ops.maxpool_fwd_multi(y0)
        ctx.update(c0=c0, y0=y0, st0=st0, idx=[p[1] for p in pooled])
        feats = [[y] for y in y0]
        cur = [p[0] for p in pooled]
        for si in range(len(self.R[0].stages)):
            for bi in range(len(self.R[0].stages[si])):
                units = [r.stages[si][bi][0] for r in self.R]          # per lane: [(ConvLayer, bn)]
                dss = [r.stages[si][bi][1] for r in self.R]
                ds = None if dss[0] is None else dss
                nu = len(units[0])
                bctx = {"x": cur, "u": []}
                # pro: inp is a raw conv output, (BnState, relu) still to be applied — what the weight gradient and the
                # saved context carry; fin: + (sums, BatchNorm tensors, count, track) for the forward launch that
                # finalises the statistics itself
                inp, pro, fin = cur, [None] * nl, [None] * nl
                for j in range(nu):
                    cls = [u[j][0] for u in units]
                    bns = [u[j][1] for u in units]
                    if j < nu - 1:
                        ops_ = self._ready(cls, inp)
                        shapes = [(x.shape[0],) + op.out_hw(x.shape[1], x.shape[2]) for op, x in zip(ops_, inp)]
                        if self._can_fold(bns, [u[j + 1][0] for u in units], shapes, inp, train):
                            c, st, fnext = self._unit_fwd_fold(cls, bns, inp, fin)
                            bctx["u"].append((inp, c, None, st, pro))
                            inp, pro = c, [(s, True) for s in st]
                            fin = [(s, True) + f for s, f in zip(st, fnext)]
                        else:
                            c, y, st, _ = self._unit_fwd(cls, bns, inp, train, pros=fin)
                            bctx["u"].append((inp, c, y, st, pro))
                            inp, pro, fin = y, [None] * nl, [None] * nl
                    else:
                        if ds is not None:
                            dcl = [d[0] for d in ds]
                            dbn = [d[1] for d in ds]
                            dops = self._ready(dcl, cur)
                            dbt = train and dbn[0].training
                            dG = [G if dbt else 1 for G in self.groups]
                            dstats = [self.pool.take(dop.Co_p, G) if dbt else None for dop, G in zip(dops, dG)]
                            dspecs = [dop.forward_spec(x, stats=s, stat_groups=G) for dop, x, s, G in zip(dops, cur, dstats, dG)]
                            run_specs(dspecs)
                            c_ds = [sp.out for sp in dspecs]
                            # (data parallel: exchanged together with the main branch's statistics in _unit_fwd)
                            c, y, st, st2 = self._unit_fwd(cls, bns, inp, train, ds_c=c_ds,
                                                           ds_stats=(dstats if dbt else None), ds_bn=dbn, pros=fin)
                            bctx["ds"] = (c_ds, st2)
                        else:
                            c, y, st, _ = self._unit_fwd(cls, bns, inp, train, res=cur, pros=fin)
                        bctx["u"].append((inp, c, y, st, pro))
                        inp, pro, fin = y, [None] * nl, [None] * nl
                ctx["blocks"].append(bctx)
                cur = inp
            for l in range(nl):
                feats[l].append(cur[l])
        return feats, ctx

    # ------------------------------------------------------------------ backward
    def _pool_bwd(self, dev):
        return _BWD_POOLS[(dev, raw_stream(dev.index))]

    def _sums(self, cs, sts):
        """zeroed f64 [groups][SLOTS][2][C] per lane from the current stream's backward pool, back to back"""
        return [_bwd_sums(c, st) for c, st in zip(cs, sts)]

    def _bn_bwd(self, douts, ys, cs, bns, sts, relu=True, g_out=None, sums=None, pools=None):
        """sums given: douts are already ReLU-masked and the sums are accumulated (fused into the producing dgrad).
        pools = per lane (pooled gradient, argmax codes): the activation gradient is the max-pool backward of the pooled
        gradient + douts[l] (or nothing), gathered inside both passes; ys[l] may then be None (mask from the raw output).
        Returns the BatchNorm input gradients, per lane."""
        nl = self.nl
        reduced = sums is not None
        if sums is None:
            sums = self._sums(cs, sts)
        dcs = [torch.empty_like(c) for c in cs]
        calls = []
        for l in range(nl):
            c, st, bn = cs[l], sts[l], bns[l]
            calls.append(dict(dout=douts[l], y=(ys[l] if ys is not None else None), x=c, gamma=bn.weight.data, st=st, dx=dcs[l],
                              dgamma=grad_of(bn.weight), dbeta=grad_of(bn.bias), H=c.shape[1], W=c.shape[2], relu=relu,
                              g_out=(g_out[l] if g_out is not None else None), sums=sums[l], sums_zeroed=True, reduced=reduced,
                              pool=((pools[l][0], pools[l][1], bn.bias.data) if pools is not None else None)))
        # (eval-mode BatchNorm: st.count is inf — no batch-statistics terms in dx, hence no exchange of the sums either)
        sync = RT.dp is not None and sts[0].count != float("inf")
        pool = self._pool_bwd(cs[0].device)
        if sync:
            if not reduced:
                ops.bn_backward_multi(calls, phase="reduce")
            _, globs = _exchange(pool, sums, out_of_place=True)
            for cl, g in zip(calls, globs):
                cl["glob"] = g
            ops.bn_backward_multi(calls, phase="apply")
        else:
            ops.bn_backward_multi(calls)
        return dcs

    def _dgrad(self, ops_, srcs, hws, kws):
        """one data gradient per lane in one launch: kws = per lane keyword dicts of ConvOp.dgrad_spec"""
        specs = [op.dgrad_spec(src, hw[0], hw[1], **kw) for op, src, hw, kw in zip(ops_, srcs, hws, kws)]
        run_specs(specs)
        return [sp.out for sp in specs]

    def _param_grads(self, cls, ops_, dcs, xs, pros=None):
        accumulate_param_grads_multi(cls, ops_, dcs, xs, pros if pros is not None else [None] * self.nl)

    def _block_bwd(self, units, ds, bctx, dout, extra, dout_sums=None, prev=None):
        """dout: gradient w.r.t. the block output.  dout_sums: set when dout came out of a data-gradient epilogue
        that already masked it with this block's output ReLU and accumulated the BatchNorm-backward sums.
        prev = (y, c, BnState) of the block that consumes the returned gradient (fused the same way).
        units: per lane [(ConvLayer, bn)], ds: per lane (ConvLayer, bn) or None; every tensor argument is a per-lane list.
"""
        nl = self.nl
        x = bctx["x"]
        hw_x = [(t.shape[1], t.shape[2]) for t in x]
        k = len(units[0])
        inp, c, y, st, pro_in = bctx["u"][k - 1]
        cls_last = [u[k - 1][0] for u in units]
        bn_last = [u[k - 1][1] for u in units]
        op_last = self._ready(cls_last, x)
        dc = None
        ds_fold = None
        joint = (ds is not None and RT.dp is not None and st[0].count != float("inf")
                 and bctx["ds"][1][0].count != float("inf"))
        if joint:
            # data parallel: the block's last BatchNorm and its downsample BatchNorm take the same gradient — both
            # first passes run before ONE exchange of their (adjacent) sums — of both lanes —, then both second passes
            c_ds, st2 = bctx["ds"]
            bn_d = [d[1] for d in ds]
            fused_in = dout_sums is not None
            pool = self._pool_bwd(c[0].device)
            s_m = dout_sums if fused_in else self._sums(c, st)
            s_d = self._sums(c_ds, st2)
            dc_ds = [torch.empty_like(t) for t in c_ds]
            dc = [torch.empty_like(t) for t in c]
            g = dout if fused_in else [torch.empty_like(t) for t in c]

            def call(l, dout_l, y_l, x_l, bn, st_l, dx_l, relu, sums, **kw):
                return dict(dout=dout_l, y=y_l, x=x_l, gamma=bn.weight.data, st=st_l, dx=dx_l, dgamma=kw.pop("dgamma", None),
                            dbeta=kw.pop("dbeta", None), H=x_l.shape[1], W=x_l.shape[2], relu=relu, sums=sums,
                            sums_zeroed=True, **kw)
            if not fused_in:
                ops.bn_backward_multi([call(l, dout[l], y[l], c[l], bn_last[l], st[l], dc[l], True, s_m[l])
                                       for l in range(nl)], phase="reduce")
            # (the downsample branch's sums need the masked gradient: dout with this block's ReLU mask, or the
            # already masked gradient of a fused producer)
            ops.bn_backward_multi([call(l, dout[l], None if fused_in else y[l], c_ds[l], bn_d[l], st2[l], dc_ds[l],
                                        not fused_in, s_d[l]) for l in range(nl)], phase="reduce")
            _, globs = _exchange(pool, list(s_m) + list(s_d), out_of_place=True)
            g_m, g_d = globs[:nl], globs[nl:]
            ops.bn_backward_multi([call(l, dout[l], None if fused_in else y[l], c[l], bn_last[l], st[l], dc[l],
                                        not fused_in, s_m[l], g_out=(None if fused_in else g[l]), reduced=fused_in,
                                        glob=g_m[l], dgamma=grad_of(bn_last[l].weight), dbeta=grad_of(bn_last[l].bias))
                                   for l in range(nl)], phase="apply")
            ops.bn_backward_multi([call(l, g[l], None, c_ds[l], bn_d[l], st2[l], dc_ds[l], False, s_d[l], glob=g_d[l],
                                        dgamma=grad_of(bn_d[l].weight), dbeta=grad_of(bn_d[l].bias)) for l in range(nl)],
                                  phase="apply")
        elif dout_sums is not None:
            g = dout
            dc = self._bn_bwd(dout, None, c, bn_last, st, sums=dout_sums)
        else:
            g = [torch.empty_like(t) for t in c]
            dc = self._bn_bwd(dout, y, c, bn_last, st, relu=True, g_out=g)
        if ds is not None:
            c_ds, st2 = bctx["ds"]
            dcl = [d[0] for d in ds]
            dop = self._ready(dcl, x)
            if not joint:
                dc_ds = self._bn_bwd(g, None, c_ds, [d[1] for d in ds], st2, relu=False)
            self._param_grads(dcl, dop, dc_ds, x)
            op_first = self._ready([u[0][0] for u in units], x)
            if k > 1 and all(o.can_fold_ds_dgrad(dp, t, t) for o, dp, t in zip(op_first, dop, dc_ds)):
                # the projection's data gradient rides in conv1's (fs_conv3x3_s2d, FsConvArgs.ds_src): its result is one
                # more K segment of the (even, even) class there, never a tensor; the feature gradient stays the addend
                ds_fold = [(dop[l], dc_ds[l]) for l in range(nl)]
                dres = extra if extra is not None else [None] * nl
            else:
                dres = self._dgrad(dop, dc_ds, hw_x, [dict(addend=(extra[l] if extra is not None else None)) for l in range(nl)])
        else:
            assert extra is None
            dres = g
        for j in range(k - 1, 0, -1):
            cls = [u[j][0] for u in units]
            op = self._ready(cls, x)
            xin, pro_x = inp, pro_in                 # the input of convolution j (raw + prologue where the forward folded)
            inp, c, y, st, pro_in = bctx["u"][j - 1]
            hw_in = [(t.shape[1], t.shape[2]) for t in xin]
            src, kw = dc, [dict() for _ in range(nl)]
            sums = None
            if y is None:
                # folded BatchNorm: the activation was never stored — the ReLU mask is the sign of scale * c + shift,
                # evaluated (with the BatchNorm-backward sums) in the data gradient's epilogue from c
                sums = self._sums(c, st)
                dy_prev = self._dgrad(op, src, hw_in, [dict(bn_fuse=(c[l], st[l], sums[l]), mask_bn=True, **kw[l]) for l in range(nl)])
            elif all(o.can_fuse_bn_bwd(t.shape[0], t.shape[1], t.shape[2], s.groups) for o, t, s in zip(op, xin, st)):
                sums = self._sums(c, st)
                dy_prev = self._dgrad(op, src, hw_in, [dict(mask=y[l], bn_fuse=(c[l], st[l], sums[l]), **kw[l]) for l in range(nl)])
            else:
                dy_prev = self._dgrad(op, src, hw_in, kw)
            self._param_grads(cls, op, dc, xin, pro_x)
            bn_prev = [u[j - 1][1] for u in units]
            if sums is not None:
                dc = self._bn_bwd(dy_prev, None, c, bn_prev, st, sums=sums)
            else:
                dc = self._bn_bwd(dy_prev, y, c, bn_prev, st, relu=True)
        cls = [u[0][0] for u in units]
        op = self._ready(cls, x)
        src, kw = dc, [dict() for _ in range(nl)]
        if ds_fold is not None:
            for l in range(nl):
                kw[l]["ds"] = ds_fold[l]
        if prev is not None and all(o.can_fuse_bn_bwd(t.shape[0], t.shape[1], t.shape[2], p.groups)
                                                    for o, t, p in zip(op, x, prev[2])):
            py, pc, pst = prev
            sums = self._sums(pc, pst)
            out = self._dgrad(op, src, hw_x, [dict(addend=dres[l], mask=py[l], bn_fuse=(pc[l], pst[l], sums[l]), **kw[l])
                                               for l in range(nl)]), sums
        else:
            out = self._dgrad(op, src, hw_x, [dict(addend=dres[l], **kw[l]) for l in range(nl)]), None
        self._param_grads(cls, op, dc, x)
        return out

    def backward(self, ctx, gfeats):
        """gfeats: per lane a list of 5 NHWC dense gradients (or None).  Accumulates parameter gradients."""
        nl = self.nl
        R0 = self.R[0]
        nst = len(R0.stages)
        xs = ctx["x"]
        bwd_pool_reset(xs[0].device)
        tag = "enc%d" % sum(x.shape[0] for x in xs)
        RT.mark(tag + ".bwd.start")
        last = ctx["blocks"][-1]["u"][-1][2]
        dout = [gfeats[l][nst] if gfeats[l][nst] is not None else torch.zeros_like(last[l]) for l in range(nl)]
        bi = len(ctx["blocks"])
        dsums = None
        for si in range(nst - 1, -1, -1):
            nblk = len(R0.stages[si])
            _ENCODER_END[0] = si <= 1           # (layer2, layer1 and the stem: what flush_deferred may share, see there)
            for b in range(nblk - 1, -1, -1):
                bi -= 1
                units = [r.stages[si][b][0] for r in self.R]
                dss = [r.stages[si][b][1] for r in self.R]
                ds = None if dss[0] is None else dss
                extra = None
                if b == 0 and si > 0 and any(gfeats[l][si] is not None for l in range(nl)):
                    if ds is None:
                        raise NotImplementedError("feature gradient into a block without downsample")
                    # (per lane: one lane's decoder reads this feature, the other's does not — the pose decoder only takes
                    # the last one; the shared launch's epilogue options are per problem)
                    extra = [gfeats[l][si] for l in range(nl)]
                prev = None
                if bi > 0 and not (b == 0 and any(gfeats[l][si] is not None for l in range(nl)) and ds is None):
                    pu = ctx["blocks"][bi - 1]["u"][-1]
                    # the consumer of the returned gradient is the previous block's output BatchNorm, unless a
                    # feature gradient still has to be added to it first (stage boundary without downsample)
                    prev = (pu[2], pu[1], pu[3])
                issue_advanced(xs[0].device)
                dout, dsums = self._block_bwd(units, ds, ctx["blocks"][bi], dout, extra, dout_sums=dsums, prev=prev)
            if RT.dp is not None and si >= 2:
                for r in self.R:
                    if r.m._pending == 1:
                        # (only the module's LAST pending backward of the step: an encoder that ran several training forwards —
                        # the pose encoder with RT.batch_pose_pairs = False — accumulates every call's gradients first; a slice
                        # reduced after the first backward would be reduced again with the second one's local sums on top)
                        # gradient bucket of this stage (reverse parameter order, like DDP): layer4 and layer3 carry 94 % of
                        # the encoder's parameters and finish first
                        RT.dp.partial_ready(r.m, [getattr(r.m, "layer%d" % (si + 1))])
        y0 = ctx["y0"]
        add0 = [gfeats[l][0] for l in range(nl)]
        if RT.stem_flush:
            # layer 1's weight gradients go to the companion now: they run beside the stem's memory-bound passes (pooling
            # backward, both BatchNorm-backward passes: the largest tensors of the network) instead of behind them
            flush_deferred(_current_stream())
        if ctx.get("stem_fused"):
            # no pooling-backward launch and no gradient tensor at the stem's resolution: both BatchNorm passes gather it
            # (the ReLU mask from the raw output for every lane: the sign of the forward's own expression — no read of
            # the activation where one was stored)
            dc0 = self._bn_bwd(add0, None, ctx["c0"], [r.m.bn1 for r in self.R], ctx["st0"], relu=True,
                               pools=[(dout[l], ctx["idx"][l]) for l in range(nl)])
        else:
            d0 = ops.maxpool_bwd_multi(dout, ctx["idx"], y0[0].shape[1], y0[0].shape[2], add0)
            dc0 = self._bn_bwd(d0, y0, ctx["c0"], [r.m.bn1 for r in self.R], ctx["st0"], relu=True)
        stems = [r.stem for r in self.R]
        self._param_grads(stems, self._ready(stems, xs), dc0, xs)
        RT.mark(tag + ".bwd.end")
        flush_deferred(_current_stream())
        _ENCODER_END[0] = False
        chain_ends(xs[0].device)


# ==============================================================================================
# Depth decoder (MultiChannelDepthDecoder)
# ==============================================================================================
class DepthDecoderRunner:
    def __init__(self, module):
        self.m = module
        self.up0, self.up1, self.disp = {}, {}, {}
        for i in range(4, -1, -1):
            b0 = module.convs[("upconv", i, 0)]
            b1 = module.convs[("upconv", i, 1)]
            self.up0[i] = (ConvLayer(b0.sequence[0]), b0.sequence[1])
            self.up1[i] = (ConvLayer(b1.sequence[0], valid_over_padded=True), b1.sequence[1])
        for s in module.scales:
            self.disp[s] = ConvLayer(module.convs[("dispconv", s)], valid_over_padded=True)
        # MultiChannelDepthDecoderUncertain: one more 3x3 replicate conv per scale -> sigmoid (depth_encoder.py:163-186)
        self.unc = {s: ConvLayer(module.convs[("uncertain_logz", s)], valid_over_padded=True)
                    for s in module.scales if ("uncertain_logz", s) in module.convs}
        self.pool = None
        self._border0 = None
        self.tail_pending = False      # its weight gradients (and gradient bucket) are waiting for the end of the backward

    def forward(self, feats, train, P2=None):
        """feats: 5 NHWC dense tensors.  Returns ({scale: (logits, depth, disp[, uncertain_z])}, ctx).
        P2 ([N,3,4] fp32 on the device) with module.base_fx set: focal-length depth scaling (depth_encoder.py:36-43)."""
        m = self.m
        dev, dt = feats[-1].device, feats[-1].dtype
        if self.pool is None or self.pool.buf.device != dev:
            self.pool = StatsPool(dev)
        self.pool.reset()
        if train:
            _check_train_bn(self.up0[4][1], "DepthDecoder")
        base_fx = getattr(m, "base_fx", None)
        if base_fx is None or P2 is None:
            P2, base_fx = None, None
        else:
            P2 = P2.detach().to(dev, torch.float32).contiguous()
        ctx = {"lv": {}, "feats": feats, "P2": P2, "base_fx": base_fx}
        outs = {}
        x = feats[-1]
        K = int(m.num_output_channels)
        # the heads of the lower levels feed nothing but the loss: each runs on the chain's companion stream beside the next
        # level (issued after that level's first kernel, so that the chain keeps its executor stream: flush_deferred).
        # Same box, with / without: headline 5.167 / 5.179 ms, ResNet-50 18.24 / 18.32, fisheye 6.48 / 6.53.
        aside = bool(RT.overlap and dev.type == "cuda" and RT.wgrad_streams)
        cur = _current_stream(dev) if aside else None
        ws = RT.companion_stream(dev, cur)[1] if aside else None
        put_off, used_ws = None, False

        def head(i, lv, y1p):
            cld = self.disp[i]
            opd = cld.ready(dt, dev)
            lv["logits"] = opd.forward(y1p, bias=cld.bias, out_f32=True)
            if i in self.unc:
                clu = self.unc[i]
                opu = clu.ready(dt, dev)
                lv["unc"] = ops.sigmoid_head_fwd(opu.forward(y1p, bias=clu.bias, out_f32=True))

        for i in range(4, -1, -1):
            cl0, bn0 = self.up0[i]
            op0 = cl0.ready(dt, dev)
            N, h, w, _ = x.shape
            st_a = self.pool.take(op0.Co_p) if train else None
            c0 = op0.forward(x, bias=cl0.bias, stats=st_a)
            if put_off is not None:
                ev, j, lvj, yj = put_off
                ws.wait_event(ev)
                with torch.cuda.stream(ws):
                    head(j, lvj, yj)
                put_off, used_ws = None, True
            world = _dp_stats(st_a) if train else 1
            y0 = torch.empty(N, h, w, op0.Co_p, dtype=dt, device=dev)
            s0 = ops.BnState(op0.Co_p, dev)
            ops.bn_apply(c0, st_a, bn_tensors(bn0), s0, y0, h, w, N * h * w * world, relu=True, track=train)
            skip = feats[i - 1] if (m.use_skips and i > 0) else None
            xcat = ops.upcat_pad_fwd(y0, skip)
            cl1, bn1 = self.up1[i]
            op1 = cl1.ready(dt, dev)
            H2, W2 = 2 * h, 2 * w
            st_b = self.pool.take(op1.Co_p) if train else None
            c1 = op1.forward(xcat, bias=cl1.bias, stats=st_b)
            world = _dp_stats(st_b) if train else 1
            y1p = torch.empty(N, H2 + 2, W2 + 2, op1.Co_p, dtype=dt, device=dev)
            s1 = ops.BnState(op1.Co_p, dev)
            ops.bn_apply(c1, st_b, bn_tensors(bn1), s1, y1p, H2, W2, N * H2 * W2 * world, relu=True, pad_out=True,
                         track=train)
            lv = dict(x=x, c0=c0, y0=y0, s0=s0, xcat=xcat, c1=c1, y1p=y1p, s1=s1, h=h, w=w,
                      Cs=(skip.shape[3] if skip is not None else 0))
            if i in m.scales:
                if aside and i > 0:
                    ev = torch.cuda.Event()
                    ev.record(cur)
                    put_off = (ev, i, lv, y1p)
                else:
                    head(i, lv, y1p)
            ctx["lv"][i] = lv
            x = y1p[:, 1:-1, 1:-1]
        if used_ws:
            cur.wait_stream(ws)
        sc = [i for i in range(4, -1, -1) if "logits" in ctx["lv"][i]]
        if getattr(m, "sigmoid_head", False):
            # base-class DepthDecoder (depth_encoder.py:90-111): disp = sigmoid(dispconv), depth = scale / (min_disp +
            # (max_disp - min_disp) * disp) — the sigmoid is the uncertainty head's kernel, the three element-wise ops
            # on a one-channel map stay in torch
            a, bq = 1.0 / float(m.max_depth), 1.0 / float(m.min_depth) - 1.0 / float(m.max_depth)
            scale = None if P2 is None else (P2[:, 0, 0] / float(base_fx)).view(-1, 1, 1, 1)
            for i in sc:
                lv = ctx["lv"][i]
                u = ops.sigmoid_head_fwd(lv["logits"])
                inv = torch.reciprocal(a + bq * u)
                depth = inv if scale is None else inv * scale
                lv["sig"] = (u, depth)
                outs[i] = (lv["logits"], depth, u)
            return outs, ctx
        # softmax-expectation heads of all scales in one launch (their outputs are only read by the loss)
        if sc:
            heads = ops.depth_head_fwd_multi([ctx["lv"][i]["logits"] for i in sc], m.depth_bins, K, m.min_depth,
                                             m.max_depth, P2=P2, base_fx=base_fx)
            for i, (depth, disp) in zip(sc, heads):
                lv = ctx["lv"][i]
                outs[i] = (lv["logits"], depth, disp) + ((lv["unc"],) if "unc" in lv else ())
        return outs, ctx

    def backward(self, ctx, g_depth, g_disp, g_unc=None):
        """g_depth / g_disp / g_unc: {scale: [N,1,H,W] fp32 or None}.  Returns the 5 feature gradients (NHWC)."""
        g_unc = g_unc or {}
        m = self.m
        feats = ctx["feats"]
        dev, dt = feats[-1].device, feats[-1].dtype
        K = int(m.num_output_channels)
        gfeats = [None] * 5
        bwd_pool_reset(dev)
        chain_begins(dev)
        RT.mark("ddec.bwd.start")
        tail = RT.dp is not None and RT.dp.decoder_tail and RT.overlap and dev.type == "cuda" and \
            torch._C._current_graph_task_id() >= 0
        if tail:
            _TAIL_SINK[0] = []
        try:
            return self._backward(ctx, g_depth, g_disp, g_unc, gfeats, dev, dt, K)
        finally:
            items, _TAIL_SINK[0] = _TAIL_SINK[0], None
            if items:
                defer_to_tail(items, self, dev)

    def _backward(self, ctx, g_depth, g_disp, g_unc, gfeats, dev, dt, K):
        m = self.m
        feats = ctx["feats"]

        # logit gradients of every scale that received one, in one launch at the head of the backward
        sc = [i for i in range(5) if i in m.scales and (g_depth.get(i) is not None or g_disp.get(i) is not None)]
        dls = {}
        if sc and getattr(m, "sigmoid_head", False):
            a, bq = 1.0 / float(m.max_depth), 1.0 / float(m.min_depth) - 1.0 / float(m.max_depth)
            for i in sc:
                u, depth = ctx["lv"][i]["sig"]
                du = g_disp[i] if g_disp.get(i) is not None else torch.zeros_like(u)
                if g_depth.get(i) is not None:
                    # d depth / d u = -scale * bq / (a + bq u)^2 = -bq * depth / (a + bq u)
                    du = du - g_depth[i] * depth * (bq / (a + bq * u))
                dls[i] = ops.sigmoid_head_bwd(u, du, self.disp[i].ready(dt, dev).Co_p, dt)
        elif sc:
            res = ops.depth_head_bwd_multi([ctx["lv"][i]["logits"] for i in sc], m.depth_bins,
                                           [g_depth.get(i) for i in sc], [g_disp.get(i) for i in sc], K, m.min_depth,
                                           m.max_depth, dt, P2=ctx["P2"], base_fx=ctx["base_fx"])
            dls = dict(zip(sc, res))

        def disp_grad(i):
            """padded-domain gradient of y1p_i from its dispconv (or zeros)."""
            lv = ctx["lv"][i]
            y1p = lv["y1p"]
            G = None
            if i in dls:
                cld = self.disp[i]
                opd = cld.ready(dt, dev)
                dl = dls[i]
                cld.accumulate_param_grads(opd, dl, y1p)
                G = opd.dgrad(dl, y1p.shape[1], y1p.shape[2])
            if i in self.unc and g_unc.get(i) is not None:
                clu = self.unc[i]
                opu = clu.ready(dt, dev)
                dlu = ops.sigmoid_head_bwd(lv["unc"], g_unc[i], opu.Co_p, dt)
                clu.accumulate_param_grads(opu, dlu, y1p)
                G = opu.dgrad(dlu, y1p.shape[1], y1p.shape[2], addend=G)
            if G is not None:
                return G, False
            # no head on this level (the lowest one): nothing to add the next gradient to, but the replicate-pad border must
            # read as zero (fs_bn_bwd_*, fold) — a buffer that is zeroed once and whose interior every step overwrites
            key = (tuple(y1p.shape), y1p.dtype, y1p.device)
            if self._border0 is None or self._border0[0] != key:
                self._border0 = (key, torch.zeros_like(y1p))
            return self._border0[1], True

        Gp, _ = disp_grad(0)
        for i in range(0, 5):
            issue_advanced(dev)
            lv = ctx["lv"][i]
            h, w = lv["h"], lv["w"]
            H2, W2 = 2 * h, 2 * w
            cl1, bn1 = self.up1[i]
            op1 = cl1.ready(dt, dev)
            y1_int = lv["y1p"][:, 1:-1, 1:-1]
            dc1 = _bn_bwd(Gp, y1_int, lv["c1"], bn1, lv["s1"], H2, W2, relu=True, fold=True)
            cl1.accumulate_param_grads(op1, dc1, lv["xcat"])
            dxcat = op1.dgrad(dc1, H2 + 2, W2 + 2)
            cl0, bn0 = self.up0[i]
            op0 = cl0.ready(dt, dev)
            # (the first pass of bn0's backward rides in the launch that produces its gradient: no fs_bn_bwd_reduce)
            sums0 = _bwd_sums(lv["c0"], lv["s0"])
            d_y0, d_skip = ops.upcat_pad_bwd(dxcat, h, w, op0.Co_p, lv["Cs"], bn=(lv["y0"], lv["c0"], lv["s0"], sums0))
            if i > 0 and lv["Cs"]:
                gfeats[i - 1] = d_skip
            dc0 = _bn_bwd(d_y0, None, lv["c0"], bn0, lv["s0"], h, w, sums=sums0)
            cl0.accumulate_param_grads(op0, dc0, lv["x"])
            if i < 4:
                Gp, fresh = disp_grad(i + 1)
                interior = Gp[:, 1:-1, 1:-1]
                op0.dgrad(dc0, h, w, out=interior, addend=None if fresh else interior)
            else:
                gfeats[4] = op0.dgrad(dc0, h, w)
        RT.mark("ddec.bwd.end")
        flush_deferred(_current_stream())
        return gfeats


# ==============================================================================================
# Pose decoder (+ fused pose tail / transform)
# ==============================================================================================
class PoseDecoderRunner:
    def __init__(self, module):
        self.m = module
        self.cl = [ConvLayer(module.convs["squeeze"]), ConvLayer(module.convs[("pose", 0)]),
                   ConvLayer(module.convs[("pose", 1)]), ConvLayer(module.convs[("pose", 2)])]

    def forward(self, feat, invert):
        """feat: NHWC last encoder feature.  Returns (axisangle, translation, T, ctx).
        invert may be a tuple of G flags: feat then stacks G calls of the module along N (no BatchNorm here, so
        the convolutions simply run on the stacked batch) and the outputs are G-tuples."""
        dt, dev = feat.dtype, feat.device
        acts = [feat]
        x = feat
        for j in range(3):
            op = self.cl[j].ready(dt, dev)
            x = op.forward(x, bias=self.cl[j].bias, relu=True)
            acts.append(x)
        op = self.cl[3].ready(dt, dev)
        x3 = op.forward(x, bias=self.cl[3].bias, out_f32=True)
        nf = int(self.m.num_frames_to_predict_for)
        if isinstance(invert, tuple):
            G = len(invert)
            B = x3.shape[0] // G
            outs = [ops.pose_tail_fwd(x3[g * B:(g + 1) * B], nf, invert[g]) for g in range(G)]
            aa, tr, T = (tuple(o[i] for o in outs) for i in range(3))
        else:
            aa, tr, T = ops.pose_tail_fwd(x3, nf, invert)
        return aa, tr, T, {"acts": acts, "x3": x3, "invert": invert, "nf": nf}

    def backward(self, ctx, dT):
        acts, x3 = ctx["acts"], ctx["x3"]
        dt, dev = acts[0].dtype, acts[0].device
        chain_begins(dev)
        RT.mark("pdec.bwd.start")
        if isinstance(ctx["invert"], tuple):
            G = len(ctx["invert"])
            B = x3.shape[0] // G
            d = torch.empty(x3.shape, dtype=dt, device=dev)
            for g in range(G):
                ops.pose_tail_bwd(x3[g * B:(g + 1) * B], dT[g], ctx["nf"], ctx["invert"][g], dt, out=d[g * B:(g + 1) * B])
        else:
            d = ops.pose_tail_bwd(x3, dT, ctx["nf"], ctx["invert"], dt)
        for j in range(3, -1, -1):
            op = self.cl[j].ready(dt, dev)
            xin = acts[j]
            self.cl[j].accumulate_param_grads(op, d, xin)
            # gradient w.r.t. the input activation, masked by the producing ReLU (none for the encoder feature)
            d = op.dgrad(d, xin.shape[1], xin.shape[2], mask=(xin if j > 0 else None))
        if RT.lanes:
            # the four weight gradients go to the companion now — only with the two-lane pass: with two chains the pose chain
            # is the step's tail (its encoder backward ends 0.5 ms after the depth encoder's) and the companion shares a
            # hardware queue with it: the hand-over put 0.15 ms of weight gradients in front of that chain (5.59 -> 5.52 ms)
            flush_deferred(_current_stream())
        return d
