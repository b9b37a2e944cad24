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
from ..hip.conv import ConvOp
from .runtime import RT, grad_of

STAT_SLOTS = ops.STAT_SLOTS
# BatchNorm-backward reduction pass fused into the epilogue of the data-gradient convolution that feeds it
FUSE_BN_BWD = os.environ.get("FSNET_AMD_FUSE_BN_BWD", "1") != "0"
# BatchNorm + ReLU between two convolutions of a residual block applied by the SECOND convolution while it stages its
# operand (and by its weight gradient), instead of a pass of its own: the normalised activation never reaches HBM
# (0: never; 1: wherever the kernels can; 2: only launches the 32x32-tile kernel takes)
FOLD_BN = int(os.environ.get("FSNET_AMD_BN_FOLD", "1"))
# the second pass of a BatchNorm's backward applied by the data gradient of the convolution in front of it while it stages
# dY (coefficients from the sums the previous data gradient's epilogue left; written out once for the weight gradient).
# Built, tested against the oracle and MEASURED slower (the data gradient stages two tensors, holds
# half the blocks per CU and takes +19 us where the pass it replaces took 15): off by default, same values as FOLD_BN.
FOLD_BN_BWD = int(os.environ.get("FSNET_AMD_BN_FOLD_BWD", "0"))


class StatsPool:
    """f64 scratch for BatchNorm batch statistics: one memset per network forward."""

    def __init__(self, device, capacity=1 << 19):
        self.buf = torch.zeros(capacity, dtype=torch.float64, device=device)
        self.off = 0

    def reset(self):
        if self.off:
            self.buf[:self.off].zero_()     # everything beyond the bump pointer was never handed out: still zero
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


def _run_param_grads(op, dc, x, gw, gb, nb, pro=None):
    op.wgrad(dc, x, gw, pro=pro)
    if gb is not None:
        ops.channel_sum(dc, gb, nb)


def _defer_param_grads(cur, item):
    if _CALLBACK_QUEUED[0] is not None and _CALLBACK_QUEUED[0] != torch._C._current_graph_task_id():
        for stale in _DEFERRED.values():       # leftovers of a backward pass that raised: not this pass's gradients
            stale[1].clear()
        _ACTIVE_CHAINS.clear()
    ent = _DEFERRED.get(cur.cuda_stream)
    if ent is None:
        ent = _DEFERRED[cur.cuda_stream] = (cur, [])
    ent[1].append(item)
    _ACTIVE_CHAINS.add(cur.cuda_stream)
    task = torch._C._current_graph_task_id()
    if _CALLBACK_QUEUED[0] != task:
        # runs once, when the autograd engine has executed every node of this backward pass and before it
        # synchronises the streams it used with the caller's stream.  (Keyed by the pass: a backward that died in an
        # exception never ran its callback, and must not keep the next one from queueing its own.)
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
        _CALLBACK_QUEUED[0] = task
    if len(ent[1]) >= RT.wgrad_flush:
        flush_deferred(cur)


def flush_deferred(cur=None, spread=False):
    """hand the collected weight-gradient work of chain stream `cur` (default: all chains) to its companion;
    `spread`: the batch that ends a network's backward may also use the other chains' companions"""
    for key in ([cur.cuda_stream] if cur is not None else list(_DEFERRED.keys())):
        ent = _DEFERRED.get(key)
        if ent is None or not ent[1]:
            continue
        chain, items = ent
        targets = [RT.companion_stream(chain.device, chain)[1]]
        if spread and RT.wgrad_spread and RT.is_side(chain):
            # the pose chain's last batch (its largest layers) otherwise runs serially after everything else has
            # finished: the other chains' companions are idle by then and take every other layer
            for key2 in _ACTIVE_CHAINS:         # chains of THIS backward pass only
                if key2 != key:
                    chain2 = _DEFERRED[key2][0]
                    targets.append(RT.companion_stream(chain2.device, chain2)[1])
        for k, ws in enumerate(targets):
            mine = items[k::len(targets)]
            if not mine:
                continue
            ws.wait_stream(chain)                   # one cross-stream edge per batch
            with torch.cuda.stream(ws):
                for it in mine:
                    _run_param_grads(*it)
            _PENDING_JOIN.add((chain, ws))
        _PENDING_KEEP.extend(items)
        ent[1].clear()


def _end_of_backward():
    _CALLBACK_QUEUED[0] = None
    flush_deferred()
    _ACTIVE_CHAINS.clear()
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


def join_companions_final():
    """the current stream waits for every companion with work in flight (before the optimizer reads gradients)"""
    flush_deferred()
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
        # (data parallel: inline on the chain stream, so that one event after a stage's last weight gradient covers
        # the arena slice its gradient bucket reduces — also inside a captured step)
        mode = RT.wgrad_streams if (dc.is_cuda and RT.overlap and RT.dp is None) else 0
        if mode:
            _defer_param_grads(_current_stream(dc.device), item)
            return
        _run_param_grads(*item)


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
    """one memset per network backward for all its BatchNorm backward sums"""
    key = (device, raw_stream(device.index))
    p = _BWD_POOLS.get(key)
    if p is None:
        p = _BWD_POOLS[key] = StatsPool(device)
    p.reset()
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
class ResNetRunner:
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

    # ------------------------------------------------------------------ forward
    def _unit_fwd_fold(self, cl, bn, x, pro=None):
        """convolution + the batch statistics of its BatchNorm only (training mode): returns (raw output, BnState, what
        the consumer needs to finalise it) — the consuming convolution derives scale / shift from the sums in its own
        prologue, applies scale * c + shift and the ReLU while staging, and its block 0 fills the BnState (mean, invstd,
        scale, shift: read by the backward) and updates the running statistics: no launch between the two convolutions"""
        op = cl.ready(x.dtype, x.device)
        N, H, W, _ = x.shape
        Ho, Wo = op.out_hw(H, W)
        G = self.groups
        stats = self.pool.take(op.Co_p, G)
        c = op.forward(x, stats=stats, stat_groups=G, pro=pro)
        world = _dp_stats(stats)
        st = ops.BnState(op.Co_p, x.device, G, affine=True)
        st.count = float((N // G) * Ho * Wo * world)
        return c, st, (stats, bn_tensors(bn), st.count, True)

    def _can_fold(self, bn, nxt_cl, c_shape, dtype, device, train):
        """unit -> next unit of a block: may the BatchNorm + ReLU in between be folded into the next convolution?"""
        if not (FOLD_BN and FUSE_BN_BWD and train and bn.training):
            return False
        N, H, W = c_shape
        nop = nxt_cl.ready(dtype, device)
        if not (nop.can_fold_input(N, H, W) and nop.can_fuse_bn_bwd(N, H, W, self.groups) and N * H * W >= 1):
            return False
        return int(FOLD_BN) != 2 or nop.plan_3x3(N, H, W, forward=True, pro_mode=1)["kernel"] == "t32"

    def _unit_fwd(self, cl, bn, x, train, relu=True, res=None, ds_c=None, ds_stats=None, ds_bn=None, pro=None):
        op = cl.ready(x.dtype, x.device)
        N, H, W, _ = x.shape
        Ho, Wo = op.out_hw(H, W)
        # a BatchNorm in eval mode inside a training step (norm_eval / frozen stages, resnet.py:169-197): running
        # statistics, no update of them, one statistics group; its backward is the batch-statistics formula with the
        # mean terms switched off (count = inf), dgamma / dbeta from the same sums
        bt = train and bn.training
        assert ds_bn is None or (train and ds_bn.training) == bt, "main and downsample BatchNorm modes differ"
        G = self.groups if bt else 1
        stats = self.pool.take(op.Co_p, G) if bt else None
        c = op.forward(x, stats=stats, stat_groups=G, pro=pro)
        if bt and ds_stats is not None and RT.dp is not None:
            # block end with a downsample branch: its statistics sit right before this conv's in the pool (taken
            # back to back) and are consumed by the same bn_apply — one exchange for both
            both = self.pool.span(ds_stats, stats)
            if both is not None:
                RT.dp.allreduce_small(both)
                world = RT.dp.world
            else:
                _dp_stats(ds_stats)
                world = _dp_stats(stats)
        else:
            world = _dp_stats(stats) if bt else 1
        y = torch.empty(N, Ho, Wo, op.Co_p, dtype=x.dtype, device=x.device)
        st = ops.BnState(op.Co_p, x.device, G)
        st2 = ops.BnState(op.Co_p, x.device, G) if ds_bn is not None else None
        count = (N // G) * Ho * Wo * world if (bt or not train) else float("inf")
        ops.bn_apply(c, stats, bn_tensors(bn), st, y, Ho, Wo, count, relu=relu,
                     res=(ds_c if ds_bn is not None else res), stats2=(ds_stats if bt else None),
                     bn2=(bn_tensors(ds_bn) if ds_bn is not None else None), st2=st2, track=bt, groups=G)
        return c, y, st, st2

    def forward(self, x, train, groups=1):
        """x: NHWC [N,H,W,Ci_p] in the compute dtype.  Returns (features NHWC x5, ctx).
        groups G > 1 (training): x stacks G independent calls of the module along N — BatchNorm statistics,
        running-statistic updates and gradients are those of G separate calls in that order, the launches are
        shared (the pose encoder's two image pairs run as one pass)."""
        if self.pool is None or self.pool.buf.device != x.device:
            self.pool = StatsPool(x.device)
        self.pool.reset()
        self.groups = groups if train else 1
        assert x.shape[0] % self.groups == 0
        ctx = {"x": x, "blocks": []}
        c0, y0, st0, _ = self._unit_fwd(self.stem, self.m.bn1, x, train)
        pooled, idx = ops.maxpool_fwd(y0)
        ctx.update(c0=c0, y0=y0, st0=st0, idx=idx)
        feats = [y0]
        cur = pooled
        for blocks in self.stages:
            for units, ds in blocks:
                bctx = {"x": cur, "u": []}
                # pro: inp is a raw conv output, (BnState, relu) still to be applied — what the weight gradient and the
                # saved context carry; fin: + (sums, BatchNorm tensors, count, track) for the forward launch that
                # finalises the statistics itself
                inp, pro, fin = cur, None, None
                for j, (cl, bn) in enumerate(units):
                    if j < len(units) - 1:
                        op = cl.ready(inp.dtype, inp.device)
                        Ho, Wo = op.out_hw(inp.shape[1], inp.shape[2])
                        if self._can_fold(bn, units[j + 1][0], (inp.shape[0], Ho, Wo), inp.dtype, inp.device, train):
                            c, st, fnext = self._unit_fwd_fold(cl, bn, inp, pro=fin)
                            bctx["u"].append((inp, c, None, st, pro))
                            inp, pro, fin = c, (st, True), (st, True) + fnext
                        else:
                            c, y, st, _ = self._unit_fwd(cl, bn, inp, train, pro=fin)
                            bctx["u"].append((inp, c, y, st, pro))
                            inp, pro, fin = y, None, None
                    else:
                        if ds is not None:
                            dop = ds[0].ready(cur.dtype, cur.device)
                            dbt = train and ds[1].training
                            dstats = self.pool.take(dop.Co_p, self.groups) if dbt else None
                            c_ds = dop.forward(cur, stats=dstats, stat_groups=(self.groups if dbt else 1))
                            # (data parallel: exchanged together with the main branch's statistics in _unit_fwd)
                            c, y, st, st2 = self._unit_fwd(cl, bn, inp, train, ds_c=c_ds, ds_stats=dstats, ds_bn=ds[1], pro=fin)
                            bctx["ds"] = (c_ds, st2)
                        else:
                            c, y, st, _ = self._unit_fwd(cl, bn, inp, train, res=cur, pro=fin)
                        bctx["u"].append((inp, c, y, st, pro))
                        inp, pro, fin = y, None, None
                ctx["blocks"].append(bctx)
                cur = inp
            feats.append(cur)
        return feats, ctx

    # ------------------------------------------------------------------ backward
    def _pend(self, g, c, st, bn, sums):
        """a BatchNorm backward whose second pass is left to the data gradient of the convolution in front of it: g = the
        masked gradient w.r.t. the BatchNorm output, sums = (sum g, sum g*xhat) of the local shard.  Data parallel: the
        global sums are exchanged here, dgamma / dbeta come from the local ones."""
        glob, local = sums, None
        if RT.dp is not None and st.count != float("inf"):
            glob = torch.empty_like(sums)
            RT.dp.allreduce_small(sums, out=glob)
            local = sums
        return dict(g=g, c=c, st=st, bn=bn, sums=glob, sums_local=local)

    @staticmethod
    def _pend_kw(pend):
        """ConvOp.dgrad arguments of a pending BatchNorm backward: (source gradient, kwargs, the tensor that receives
        the BatchNorm input gradient)"""
        dc = torch.empty_like(pend["c"])
        bn = pend["bn"]
        return pend["g"], dict(pro_bwd=dict(c=pend["c"], st=pend["st"], gamma=bn.weight.data, sums=pend["sums"],
                                            sums_local=pend["sums_local"], dgamma=grad_of(bn.weight),
                                            dbeta=grad_of(bn.bias), dc_out=dc)), dc

    def _can_fold_bwd(self, op, bn, st, c):
        if not (FOLD_BN_BWD and FUSE_BN_BWD and st.count != float("inf")
                and op.can_fold_bn_bwd(c.shape[0], c.shape[1], c.shape[2])):
            return False
        return int(FOLD_BN_BWD) != 2 or op.plan_3x3(c.shape[0], c.shape[1], c.shape[2], forward=False, pro_mode=2)["kernel"] == "t32"

    def _block_bwd(self, units, ds, bctx, dout, extra, dout_sums=None, prev=None):
        """dout: gradient w.r.t. the block output.  dout_sums: set when dout came out of a data-gradient epilogue
        that already masked it with this block's output ReLU and accumulated the BatchNorm-backward sums.
        prev = (y, c, BnState) of the block that consumes the returned gradient (fused the same way).

        A BatchNorm whose masked output gradient and backward sums exist (they come out of the epilogue of the data
        gradient behind it) does not get a second pass of its own where the convolution in front of it is a 3x3 /
        stride-1 one: that convolution's data gradient applies dx = gamma*invstd*(g - mean_g - xhat*mean_gx) to dY while it
        stages it (`pend`), writes it out once, and the weight gradient reads that."""
        x = bctx["x"]
        N, H, W, _ = x.shape
        k = len(units)
        inp, c, y, st, pro_in = bctx["u"][k - 1]
        Ho, Wo = y.shape[1], y.shape[2]
        op_last = units[k - 1][0].ready(x.dtype, x.device)
        bn_last = units[k - 1][1]
        fold_last = dout_sums is not None and self._can_fold_bwd(op_last, bn_last, st, c)
        pend, dc = None, None
        joint = (ds is not None and RT.dp is not None and st.count != float("inf")
                 and bctx["ds"][1].count != float("inf"))
        if joint:
            # data parallel: the block's last BatchNorm and its downsample BatchNorm take the same gradient — both
            # first passes run before ONE exchange of their (adjacent) sums, then both second passes
            c_ds, st2 = bctx["ds"]
            dop = ds[0].ready(x.dtype, x.device)
            bn_m, bn_d = bn_last, ds[1]
            fused_in = dout_sums is not None
            pool = _BWD_POOLS[(c.device, raw_stream(c.device.index))]
            s_m = dout_sums if fused_in else _bwd_sums(c, st)
            s_d = _bwd_sums(c_ds, st2)
            dc_ds = torch.empty_like(c_ds)
            if not fold_last:
                dc = torch.empty_like(c)
            g = dout if fused_in else torch.empty_like(c)
            if not fused_in:
                ops.bn_backward(dout, y, c, bn_m.weight.data, st, dc, None, None, Ho, Wo, relu=True, sums=s_m,
                                sums_zeroed=True, phase="reduce")
            # (the downsample branch's sums need the masked gradient: dout with this block's ReLU mask, or the
            # already masked gradient of a fused producer)
            ops.bn_backward(dout, None if fused_in else y, c_ds, bn_d.weight.data, st2, dc_ds, None, None, Ho, Wo,
                            relu=not fused_in, sums=s_d, sums_zeroed=True, phase="reduce")
            both = pool.span(s_m, s_d)
            if both is not None:
                glob = torch.empty_like(both)
                RT.dp.allreduce_small(both, out=glob)
                g_m, g_d = glob[: s_m.numel()].view(s_m.shape), glob[s_m.numel():].view(s_d.shape)
            else:
                g_m, g_d = torch.empty_like(s_m), torch.empty_like(s_d)
                RT.dp.allreduce_small(s_m, out=g_m)
                RT.dp.allreduce_small(s_d, out=g_d)
            if fold_last:
                pend = dict(g=g, c=c, st=st, bn=bn_m, sums=g_m, sums_local=s_m)
            else:
                ops.bn_backward(dout, None if fused_in else y, c, bn_m.weight.data, st, dc, grad_of(bn_m.weight),
                                grad_of(bn_m.bias), Ho, Wo, relu=not fused_in, g_out=(None if fused_in else g), sums=s_m,
                                sums_zeroed=True, reduced=fused_in, phase="apply", glob=g_m)
            ops.bn_backward(g, None, c_ds, bn_d.weight.data, st2, dc_ds, grad_of(bn_d.weight), grad_of(bn_d.bias), Ho, Wo,
                            relu=False, sums=s_d, sums_zeroed=True, phase="apply", glob=g_d)
        elif dout_sums is not None:
            g = dout
            if fold_last:
                pend = self._pend(g, c, st, bn_last, dout_sums)
            else:
                dc = _bn_bwd(dout, None, c, bn_last, st, Ho, Wo, sums=dout_sums)
        else:
            g = torch.empty_like(c)
            dc = _bn_bwd(dout, y, c, bn_last, st, Ho, Wo, relu=True, g_out=g)
        if ds is not None:
            c_ds, st2 = bctx["ds"]
            dop = ds[0].ready(x.dtype, x.device)
            if not joint:
                dc_ds = _bn_bwd(g, None, c_ds, ds[1], st2, Ho, Wo, relu=False)
            ds[0].accumulate_param_grads(dop, dc_ds, x)
            dres = dop.dgrad(dc_ds, H, W, addend=extra)
        else:
            assert extra is None
            dres = g
        for j in range(k - 1, 0, -1):
            cl = units[j][0]
            op = cl.ready(x.dtype, x.device)
            xin, pro_x = inp, pro_in                 # the input of convolution j (raw + prologue where the forward folded)
            inp, c, y, st, pro_in = bctx["u"][j - 1]
            src, kw = dc, {}
            if pend is not None:
                src, kw, dc = self._pend_kw(pend)
            sums = None
            if y is None:
                # folded BatchNorm: the activation was never stored — the ReLU mask is the sign of scale * c + shift,
                # evaluated (with the BatchNorm-backward sums) in the data gradient's epilogue from c
                sums = _bwd_sums(c, st)
                dy_prev = op.dgrad(src, xin.shape[1], xin.shape[2], bn_fuse=(c, st, sums), mask_bn=True, **kw)
            elif FUSE_BN_BWD and op.can_fuse_bn_bwd(N, xin.shape[1], xin.shape[2], st.groups):
                sums = _bwd_sums(c, st)
                dy_prev = op.dgrad(src, xin.shape[1], xin.shape[2], mask=y, bn_fuse=(c, st, sums), **kw)
            else:
                dy_prev = op.dgrad(src, xin.shape[1], xin.shape[2], **kw)
            # (after the data gradient: with a pending BatchNorm backward, dc is its side output)
            cl.accumulate_param_grads(op, dc, xin, pro=pro_x)
            bn_prev = units[j - 1][1]
            pend = None
            if sums is not None and self._can_fold_bwd(units[j - 1][0].ready(x.dtype, x.device), bn_prev, st, c):
                pend = self._pend(dy_prev, c, st, bn_prev, sums)
            elif sums is not None:
                dc = _bn_bwd(dy_prev, None, c, bn_prev, st, c.shape[1], c.shape[2], sums=sums)
            else:
                dc = _bn_bwd(dy_prev, y, c, bn_prev, st, y.shape[1], y.shape[2], relu=True)
        cl = units[0][0]
        op = cl.ready(x.dtype, x.device)
        src, kw = dc, {}
        if pend is not None:
            src, kw, dc = self._pend_kw(pend)
        if prev is not None and FUSE_BN_BWD and op.can_fuse_bn_bwd(N, H, W, prev[2].groups):
            py, pc, pst = prev
            sums = _bwd_sums(pc, pst)
            out = op.dgrad(src, H, W, addend=dres, mask=py, bn_fuse=(pc, pst, sums), **kw), sums
        else:
            out = op.dgrad(src, H, W, addend=dres, **kw), None
        cl.accumulate_param_grads(op, dc, x)
        return out

    def backward(self, ctx, gfeats):
        """gfeats: list of 5 NHWC dense gradients (or None).  Accumulates parameter gradients."""
        nst = len(self.stages)
        bwd_pool_reset(ctx["x"].device)
        tag = "enc%d" % ctx["x"].shape[0]
        RT.mark(tag + ".bwd.start")
        dout = gfeats[nst]
        last = ctx["blocks"][-1]["u"][-1][2]
        if dout is None:
            dout = torch.zeros_like(last)
        bi = len(ctx["blocks"])
        dsums = None
        for si in range(nst - 1, -1, -1):
            blocks = self.stages[si]
            for b in range(len(blocks) - 1, -1, -1):
                bi -= 1
                units, ds = blocks[b]
                extra = gfeats[si] if (b == 0 and si > 0) else None
                if extra is not None and ds is None:
                    raise NotImplementedError("feature gradient into a block without downsample")
                prev = None
                if bi > 0 and not (b == 0 and gfeats[si] is not None and ds is None):
                    pu = ctx["blocks"][bi - 1]["u"][-1]
                    # the consumer of the returned gradient is the previous block's output BatchNorm, unless a
                    # feature gradient still has to be added to it first (stage boundary without downsample)
                    prev = (pu[2], pu[1], pu[3])
                dout, dsums = self._block_bwd(units, ds, ctx["blocks"][bi], dout, extra, dout_sums=dsums, prev=prev)
            if RT.dp is not None and si >= 2 and self.m._pending == 1:
                # (only the module's LAST pending backward of the step: an encoder that ran several training forwards —
                # the pose encoder with FSNET_AMD_BATCH_POSE=0 — accumulates every call's gradients first; a slice
                # reduced after the first backward would be reduced again with the second one's local sums on top)
                # gradient bucket of this stage (reverse parameter order, like DDP): layer4 and layer3 carry 94 % of
                # the encoder's parameters and finish first
                RT.dp.partial_ready(self.m, [getattr(self.m, "layer%d" % (si + 1))])
        y0 = ctx["y0"]
        d0 = ops.maxpool_bwd(dout, ctx["idx"], y0.shape[1], y0.shape[2], addend=gfeats[0])
        dc0 = _bn_bwd(d0, y0, ctx["c0"], self.m.bn1, ctx["st0"], y0.shape[1], y0.shape[2], relu=True)
        op = self.stem.ready(y0.dtype, y0.device)
        self.stem.accumulate_param_grads(op, dc0, ctx["x"])
        RT.mark(tag + ".bwd.end")
        flush_deferred(_current_stream(), spread=True)


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
        for i in range(4, -1, -1):
            cl0, bn0 = self.up0[i]
            op0 = cl0.ready(dt, dev)
            N, h, w, _ = x.shape
            st_a = self.pool.take(op0.Co_p) if train else None
            c0 = op0.forward(x, bias=cl0.bias, stats=st_a)
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
                cld = self.disp[i]
                opd = cld.ready(dt, dev)
                logits = opd.forward(y1p, bias=cld.bias, out_f32=True)
                lv["logits"] = logits
                if i in self.unc:
                    clu = self.unc[i]
                    opu = clu.ready(dt, dev)
                    lv["unc"] = ops.sigmoid_head_fwd(opu.forward(y1p, bias=clu.bias, out_f32=True))
            ctx["lv"][i] = lv
            x = y1p[:, 1:-1, 1:-1]
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
        RT.mark("ddec.bwd.start")

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
            return G if G is not None else torch.zeros_like(y1p)

        Gp = disp_grad(0)
        for i in range(0, 5):
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
            d_y0, d_skip = ops.upcat_pad_bwd(dxcat, h, w, op0.Co_p, lv["Cs"])
            if i > 0 and lv["Cs"]:
                gfeats[i - 1] = d_skip
            dc0 = _bn_bwd(d_y0, lv["y0"], lv["c0"], bn0, lv["s0"], h, w, relu=True)
            cl0.accumulate_param_grads(op0, dc0, lv["x"])
            if i < 4:
                Gp = disp_grad(i + 1)
                interior = Gp[:, 1:-1, 1:-1]
                op0.dgrad(dc0, h, w, out=interior, addend=interior)
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
        flush_deferred(_current_stream())
        return d
