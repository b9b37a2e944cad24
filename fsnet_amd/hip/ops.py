"""Thin host wrappers (ctypes -> libfsnet_hip.so) for the non-conv kernels.  torch tensors are used
only as device memory + stream ordering; every arithmetic step runs in the HIP library."""
import ctypes as C
import os

import torch

from .binding import (lib, check, stream_ptr, FsBnApplyArgs, FsBnBwdArgs, FsPhotoArgs, FsSmoothArgs)
from .conv import dtype_code, _timed, BN_EPS, BN_MOMENTUM, Spec, run_specs

STAT_SLOTS = 8   # FS_STAT_SLOTS in include/fsnet_hip.h


def _p(t):
    return t.data_ptr() if t is not None else None


def nchw_to_nhwc(a, b, cp, dtype, out=None):
    """a (and optionally b, concatenated along C): NCHW fp32 -> NHWC [N,H,W,cp] in `dtype`."""
    a = a.contiguous()
    N, Ca, H, W = a.shape
    Cb = 0
    if b is not None:
        b = b.contiguous()
        Cb = b.shape[1]
    if out is None:
        out = torch.empty(N, H, W, cp, dtype=dtype, device=a.device)
    assert out.shape == (N, H, W, cp) and out.is_contiguous() and out.dtype == dtype
    check(lib.fs_nchw_to_nhwc(a.data_ptr(), _p(b), out.data_ptr(), N, Ca, Cb, H, W, cp, dtype_code(dtype),
                              stream_ptr()), "nchw_to_nhwc")
    return out


class BnState:
    """Device-side state of one BatchNorm invocation (saved for backward)."""
    __slots__ = ("mean", "invstd", "count", "groups", "scale", "shift")

    def __init__(self, C, device, groups=1, affine=False):
        self.mean = torch.empty(groups * C, dtype=torch.float32, device=device)
        self.invstd = torch.empty(groups * C, dtype=torch.float32, device=device)
        self.count = 0.0          # rows per statistics group (x world size)
        self.groups = groups
        # folded BatchNorm (bn_finalize): y = scale * x + shift per (group, channel), applied by the consumer
        self.scale = torch.empty(groups * C, dtype=torch.float32, device=device) if affine else None
        self.shift = torch.empty(groups * C, dtype=torch.float32, device=device) if affine else None


def bn_finalize(stats, bn, st, Cc, count, track=True, groups=1):
    """the statistics half of bn_apply (fs_bn_finalize): fills st.mean / invstd / scale / shift ([groups][C]) and
    updates the running statistics; the activation is normalised by the kernels that read it."""
    a = FsBnApplyArgs()
    a.groups = groups
    assert st.groups == groups and st.scale is not None
    a.stats = _p(stats)
    a.gamma, a.beta = bn["weight"].data_ptr(), bn["bias"].data_ptr()
    if track or stats is None:
        a.running_mean, a.running_var = bn["running_mean"].data_ptr(), bn["running_var"].data_ptr()
        a.num_batches_tracked = bn["num_batches_tracked"].data_ptr() if track else None
    a.save_mean, a.save_invstd = st.mean.data_ptr(), st.invstd.data_ptr()
    st.count = float(count)
    a.count, a.eps, a.momentum = float(count), BN_EPS, BN_MOMENTUM
    a.C = Cc
    check(lib.fs_bn_finalize(C.byref(a), st.scale.data_ptr(), st.shift.data_ptr(), stream_ptr()), "bn_finalize")
    return st


def bn_apply(x, stats, bn, st, y, H, W, count, **kw):
    """one BatchNorm (+ residual, + ReLU) forward pass, issued now (see bn_apply_spec)"""
    run_specs([bn_apply_spec(x, stats, bn, st, y, H, W, count, **kw)])
    return y


def bn_apply_spec(x, stats, bn, st, y, H, W, count, relu=True, pad_out=False, res=None, stats2=None, bn2=None, st2=None,
                  track=True, groups=1, pool=None):
    """-> Spec (conv.run_specs issues it, alone or sharing a launch with another network's BatchNorm of the same width).
    x: dense [N,H,W,C] raw conv output.  bn/bn2: dict(weight,bias,running_mean,running_var,nbt).
    groups G > 1: G stacked invocations of the module (count is per group; stats are [G][SLOTS][2][C])."""
    a = FsBnApplyArgs()
    a.groups = groups
    assert st.groups == groups and (st2 is None or st2.groups == groups)
    Cc = x.shape[-1]
    a.x, a.res, a.y = x.data_ptr(), _p(res), _p(y)
    if pool is not None:
        # pool = (pooled [N,H/2,W/2,C], argmax codes uint8): the stem's MaxPool2d(3, 2, 1) in the same pass; y may be None
        assert pool[0].is_contiguous() and pool[1].is_contiguous() and pool[0].shape == (x.shape[0], H // 2, W // 2, Cc)
        assert stats is not None and res is None and bn2 is None and not pad_out and H % 2 == 0 and W % 2 == 0
        a.pool_y, a.pool_idx = pool[0].data_ptr(), pool[1].data_ptr()
    else:
        assert y is not None
    a.stats, a.stats2 = _p(stats), _p(stats2)
    a.gamma, a.beta = bn["weight"].data_ptr(), bn["bias"].data_ptr()
    if track or stats is None:
        a.running_mean, a.running_var = bn["running_mean"].data_ptr(), bn["running_var"].data_ptr()
        a.num_batches_tracked = bn["num_batches_tracked"].data_ptr()
    a.save_mean, a.save_invstd = st.mean.data_ptr(), st.invstd.data_ptr()
    st.count = float(count)
    if bn2 is not None:
        a.gamma2, a.beta2 = bn2["weight"].data_ptr(), bn2["bias"].data_ptr()
        if track or stats2 is None:
            a.running_mean2, a.running_var2 = bn2["running_mean"].data_ptr(), bn2["running_var"].data_ptr()
            a.num_batches_tracked2 = bn2["num_batches_tracked"].data_ptr() if track else None
        a.save_mean2, a.save_invstd2 = st2.mean.data_ptr(), st2.invstd.data_ptr()
        st2.count = float(count)
    a.count, a.eps, a.momentum = float(count), BN_EPS, BN_MOMENTUM
    if pad_out:
        assert y.shape[1] == H + 2 and y.shape[2] == W + 2
    if y is not None:
        a.yN, a.yH, a.yW = y.stride(0), y.stride(1), y.stride(2)
    a.M, a.C, a.H, a.W = x.shape[0] * H * W, Cc, H, W
    a.relu, a.pad_out = int(relu), int(pad_out)
    nb = x.numel() * x.element_size() * (1 + (y is not None) + (res is not None))
    if pool is not None:
        nb += pool[0].numel() * pool[0].element_size() + pool[1].numel()
    return Spec(["bn_apply"], a, dtype_code(x.dtype), "bn_apply_pool" if pool is not None else "bn_apply", nb,
                lambda: "[%d,%d,%d,%d]%s%s%s" % (x.shape[0], H, W, Cc, " +res" if res is not None else "",
                                                 " pad" if pad_out else "", " +pool" if pool is not None else ""),
                y, "bn_apply")


def bn_backward(dout, y, x, gamma, st, dx, dgamma, dbeta, H, W, relu=True, fold=False, g_out=None, sums=None,
                allreduce=None, sums_zeroed=False, reduced=False, phase="all", glob=None, pool=None):
    """Two-pass BN backward.  dout: grad w.r.t. the block output (strided view, or the padded buffer
    when fold=True); y: saved output activation (interior view) for the ReLU mask.
    reduced=True: the first pass already happened in the epilogue of the convolution that produced dout
    (ConvOp.dgrad(bn_fuse=...)): dout is ReLU-masked and `sums` holds (sum g, sum g*xhat)."""
    bn_backward_multi([dict(dout=dout, y=y, x=x, gamma=gamma, st=st, dx=dx, dgamma=dgamma, dbeta=dbeta, H=H, W=W, relu=relu,
                            fold=fold, g_out=g_out, sums=sums, sums_zeroed=sums_zeroed, reduced=reduced, glob=glob, pool=pool)],
                      allreduce=allreduce, phase=phase)
    return None if phase == "reduce" else dx


def _bn_bwd_call(dout, y, x, gamma, st, dx, dgamma, dbeta, H, W, relu=True, fold=False, g_out=None, sums=None,
                 sums_zeroed=False, reduced=False, glob=None, pool=None):
    """argument struct of one BatchNorm backward -> dict(a=, code=, nb=, shp=, sums=, reduced=, glob=, y=, g_out=).
    pool = (pooled gradient [N,H/2,W/2,C], argmax codes, beta): the gradient w.r.t. the activation is the max-pool backward
    of the pooled gradient (+ dout, which may be None) gathered inside both passes (FsBnBwdArgs.pool_dy); with y None the
    ReLU mask is the sign of the forward's own scale * x + shift."""
    if reduced:
        assert sums is not None and not fold and g_out is None
        relu, y = False, None
    Cc = x.shape[-1]
    a = FsBnBwdArgs()
    if sums is None:
        sums = torch.zeros(st.groups * STAT_SLOTS, 2, Cc, dtype=torch.float64, device=x.device)
    elif not sums_zeroed and not reduced:
        sums.zero_()
    a.dout, a.y, a.x, a.dx, a.g_out = _p(dout), _p(y), x.data_ptr(), _p(dx), _p(g_out)
    if pool is not None:
        assert not reduced and not fold and g_out is None and pool[0].is_contiguous() and pool[1].is_contiguous()
        assert (dout is None or dout.is_contiguous()) and (y is None or y.is_contiguous())
        a.pool_dy, a.pool_idx, a.beta = pool[0].data_ptr(), pool[1].data_ptr(), pool[2].data_ptr()
    else:
        assert dout is not None
    a.sums = sums.data_ptr()
    a.gamma, a.save_mean, a.save_invstd = gamma.data_ptr(), st.mean.data_ptr(), st.invstd.data_ptr()
    a.dgamma, a.dbeta = _p(dgamma), _p(dbeta)
    a.count = st.count
    a.groups = st.groups
    if dout is not None:
        a.gN, a.gH, a.gW = dout.stride(0), dout.stride(1), dout.stride(2)
    if y is not None:
        a.yN, a.yH, a.yW = y.stride(0), y.stride(1), y.stride(2)
    a.M, a.C, a.H, a.W = x.shape[0] * H * W, Cc, H, W
    a.relu, a.fold = int(relu), int(fold)
    shape = (x.shape[0], H, W, Cc)
    return dict(a=a, code=dtype_code(x.dtype), nb=x.numel() * x.element_size(), sums=sums, reduced=reduced, glob=glob,
                has_y=y is not None, has_g=g_out is not None, has_dout=dout is not None,
                shp=lambda: "[%d,%d,%d,%d]%s%s" % (shape + (" fold" if fold else "", " pool" if pool is not None else "")))


def bn_backward_multi(calls, allreduce=None, phase="all", span=None):
    """The BatchNorm backwards of one or two networks' layers of the same width; each pass is ONE launch for all of them
    (fs_bn_bwd_reduce2 / fs_bn_bwd_apply2).  calls: keyword dicts of bn_backward (above) without allreduce / phase.
    allreduce(sums, out=): SyncBN exchange of the sums between the passes — dgamma / dbeta come from the local sums, dx
    from the global ones; span(a, b) -> one contiguous view over two sums buffers taken back to back (or None): the
    exchange of two lanes is then one collective.
    phase "reduce": only the first pass (the caller exchanges the sums of several BatchNorms in one collective and comes
    back with phase "apply", glob = the exchanged sums in each call; `sums` then holds the local ones)."""
    cs = [_bn_bwd_call(**c) for c in calls]
    if phase != "apply":
        run_specs([Spec(["bn_bwd_reduce"], c["a"], c["code"], "bn_bwd_reduce", c["nb"] * (1 + c["has_dout"] + c["has_y"]), c["shp"], None,
                        "bn_bwd_reduce") for c in cs if not c["reduced"]])
    if phase == "reduce":
        return
    pending = [c for c in cs if c["glob"] is None] if allreduce is not None else []
    if len(pending) == 2 and span is not None:
        both = span(pending[0]["sums"], pending[1]["sums"])
        if both is not None:
            out = torch.empty_like(both)
            allreduce(both, out=out)
            n0 = pending[0]["sums"].numel()
            pending[0]["glob"] = out[:n0].view(pending[0]["sums"].shape)
            pending[1]["glob"] = out[n0:].view(pending[1]["sums"].shape)
            pending = []
    for c in pending:
        # SyncBN: reduce out of place instead of cloning the local copy first (one device copy per BatchNorm and step)
        c["glob"] = torch.empty_like(c["sums"])
        allreduce(c["sums"], out=c["glob"])
    for c in cs:
        if c["glob"] is not None:
            c["a"].sums_local, c["a"].sums = c["sums"].data_ptr(), c["glob"].data_ptr()
    run_specs([Spec(["bn_bwd_apply"], c["a"], c["code"], "bn_bwd_apply", c["nb"] * (2 + c["has_dout"] + c["has_y"] + c["has_g"]), c["shp"],
                    None, "bn_bwd_apply") for c in cs])


def maxpool_fwd(x):
    return maxpool_fwd_multi([x])[0]


def maxpool_fwd_multi(xs):
    """[x NHWC] (one tensor, or two with the same H, W, C: ONE launch) -> [(y, idx)]"""
    outs = []
    for x in xs:
        N, H, W, Cc = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        outs.append((torch.empty(N, Ho, Wo, Cc, dtype=x.dtype, device=x.device),
                     torch.empty(N, Ho, Wo, Cc, dtype=torch.uint8, device=x.device)))
    if len(xs) == 2 and xs[0].shape[1:] == xs[1].shape[1:] and xs[0].dtype == xs[1].dtype:
        x, x1 = xs
        (y, idx), (y1, idx1) = outs
        N, H, W, Cc = x.shape
        _timed("maxpool_fwd", sum(t.numel() * t.element_size() * 1.25 + o[1].numel() for t, o in zip(xs, outs)),
               lambda: check(lib.fs_maxpool_fwd2(x.data_ptr(), y.data_ptr(), idx.data_ptr(), N, x1.data_ptr(), y1.data_ptr(),
                                                 idx1.data_ptr(), x1.shape[0], H, W, Cc, dtype_code(x.dtype), stream_ptr()),
                             "maxpool_fwd"), tag=str(tuple(x.shape)) + " || " + str(tuple(x1.shape)))
        return outs
    for x, (y, idx) in zip(xs, outs):
        N, H, W, Cc = x.shape
        _timed("maxpool_fwd", x.numel() * x.element_size() * 1.25 + idx.numel(),
               lambda: check(lib.fs_maxpool_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), N, H, W, Cc, dtype_code(x.dtype),
                                                stream_ptr()), "maxpool_fwd"), tag=str(tuple(x.shape)))
    return outs


def maxpool_bwd(dy, idx, H, W, addend=None):
    return maxpool_bwd_multi([dy], [idx], H, W, [addend])[0]


def maxpool_bwd_multi(dys, idxs, H, W, addends):
    """gradients of maxpool_fwd_multi's inputs (+ addend): one launch for two tensors of the same H, W, C"""
    dxs = [torch.empty(dy.shape[0], H, W, dy.shape[3], dtype=dy.dtype, device=dy.device) for dy in dys]
    work = lambda dx, idx, ad: dx.numel() * dx.element_size() * (1.25 + (ad is not None)) + idx.numel()
    if len(dys) == 2 and dys[0].shape[1:] == dys[1].shape[1:] and dys[0].dtype == dys[1].dtype:
        dy, dy1 = dys
        Cc = dy.shape[3]
        _timed("maxpool_bwd", sum(work(dx, i, a) for dx, i, a in zip(dxs, idxs, addends)),
               lambda: check(lib.fs_maxpool_bwd2(dy.data_ptr(), idxs[0].data_ptr(), _p(addends[0]), dxs[0].data_ptr(),
                                                 dy.shape[0], dy1.data_ptr(), idxs[1].data_ptr(), _p(addends[1]),
                                                 dxs[1].data_ptr(), dy1.shape[0], H, W, Cc, dtype_code(dy.dtype),
                                                 stream_ptr()), "maxpool_bwd"),
               tag=str(tuple(dxs[0].shape)) + " || " + str(tuple(dxs[1].shape)))
        return dxs
    for dy, idx, ad, dx in zip(dys, idxs, addends, dxs):
        _timed("maxpool_bwd", work(dx, idx, ad),
               lambda: check(lib.fs_maxpool_bwd(dy.data_ptr(), idx.data_ptr(), _p(ad), dx.data_ptr(), dy.shape[0], H, W,
                                                dy.shape[3], dtype_code(dy.dtype), stream_ptr()), "maxpool_bwd"),
               tag=str(tuple(dx.shape)))
    return dxs


def upcat_pad_fwd(a, b):
    N, h, w, Ca = a.shape
    Cb = b.shape[3] if b is not None else 0
    out = torch.empty(N, 2 * h + 2, 2 * w + 2, Ca + Cb, dtype=a.dtype, device=a.device)
    _timed("upcat_pad_fwd", (a.numel() + (b.numel() if b is not None else 0) + out.numel()) * a.element_size(),
           lambda: check(lib.fs_upcat_pad_fwd(a.data_ptr(), _p(b), out.data_ptr(), N, h, w, Ca, Cb, dtype_code(a.dtype),
                                              stream_ptr()), "upcat_pad_fwd"), tag=str(tuple(out.shape)))
    return out


def upcat_pad_bwd(dpad, h, w, Ca, Cb, bn=None):
    """bn = (y, x, BnState, sums): also the first pass of the BatchNorm backward the `da` half feeds (fs_upcat_pad_bwd_bn):
    da comes back masked by y > 0 and `sums` (zeroed f64 [SLOTS][2][Ca]) holds (sum g, sum g * xhat)"""
    N = dpad.shape[0]
    da = torch.empty(N, h, w, Ca, dtype=dpad.dtype, device=dpad.device)
    db = torch.empty(N, 2 * h, 2 * w, Cb, dtype=dpad.dtype, device=dpad.device) if Cb else None
    nbytes = (da.numel() + (db.numel() if db is not None else 0) + dpad.numel()) * da.element_size()
    if bn is not None:
        y, x, st, sums = bn
        assert y.is_contiguous() and x.is_contiguous() and y.shape == da.shape == x.shape and st.groups == 1
        assert sums.dtype == torch.float64 and sums.numel() == STAT_SLOTS * 2 * Ca
        _timed("upcat_pad_bwd", nbytes + 2 * da.numel() * da.element_size(),
               lambda: check(lib.fs_upcat_pad_bwd_bn(dpad.data_ptr(), da.data_ptr(), _p(db), N, h, w, Ca, Cb, y.data_ptr(),
                                                     x.data_ptr(), st.mean.data_ptr(), st.invstd.data_ptr(), sums.data_ptr(),
                                                     dtype_code(dpad.dtype), stream_ptr()), "upcat_pad_bwd_bn"),
               tag=str(tuple(dpad.shape)) + " bn")
        return da, db
    _timed("upcat_pad_bwd", nbytes,
           lambda: check(lib.fs_upcat_pad_bwd(dpad.data_ptr(), da.data_ptr(), _p(db), N, h, w, Ca, Cb,
                                              dtype_code(dpad.dtype), stream_ptr()), "upcat_pad_bwd"),
           tag=str(tuple(dpad.shape)))
    return da, db


def channel_sum(x, out, creal):
    """out[c] += sum over rows of dense x [..., C]."""
    Cc = x.shape[-1]
    M = x.numel() // Cc
    _timed("channel_sum", x.numel() * x.element_size(),
           lambda: check(lib.fs_channel_sum(x.data_ptr(), out.data_ptr(), M, Cc, creal, dtype_code(x.dtype),
                                            stream_ptr()), "channel_sum"), tag=str(tuple(x.shape)))


def channel_sum_multi(items):
    """[(x, out, creal)]: out[c] += column sums of dense x [..., C] — one launch per 16 (same dtype)"""
    items = list(items)
    if len(items) == 1:
        return channel_sum(*items[0])
    for i in range(0, len(items), 16):
        chunk = items[i:i + 16]
        n = len(chunk)
        code = dtype_code(chunk[0][0].dtype)
        assert all(dtype_code(x.dtype) == code for x, _, _ in chunk)
        xs = (C.c_void_p * n)(*[x.data_ptr() for x, _, _ in chunk])
        outs = (C.c_void_p * n)(*[o.data_ptr() for _, o, _ in chunk])
        Ms = (C.c_int64 * n)(*[x.numel() // x.shape[-1] for x, _, _ in chunk])
        Cs = (C.c_int32 * n)(*[x.shape[-1] for x, _, _ in chunk])
        Cr = (C.c_int32 * n)(*[int(cr) for _, _, cr in chunk])
        _timed("channel_sum", sum(x.numel() * x.element_size() for x, _, _ in chunk),
               lambda: check(lib.fs_channel_sum_multi(xs, outs, Ms, Cs, Cr, n, code, stream_ptr()), "channel_sum_multi"),
               tag="%d tensors" % n)


def depth_head_fwd(logits, bins, K, min_depth, max_depth):
    N, H, W, Cl = logits.shape
    depth = torch.empty(N, 1, H, W, dtype=torch.float32, device=logits.device)
    disp = torch.empty(N, 1, H, W, dtype=torch.float32, device=logits.device)
    check(lib.fs_depth_head_fwd(logits.data_ptr(), bins.data_ptr(), depth.data_ptr(), disp.data_ptr(), N * H * W, K,
                                Cl, float(min_depth), float(max_depth), stream_ptr()), "depth_head_fwd")
    return depth, disp


def depth_head_bwd(logits, bins, d_depth, d_disp, K, min_depth, max_depth, dtype):
    N, H, W, Cl = logits.shape
    dl = torch.empty(N, H, W, Cl, dtype=dtype, device=logits.device)
    check(lib.fs_depth_head_bwd(logits.data_ptr(), bins.data_ptr(), _p(d_depth), _p(d_disp), dl.data_ptr(),
                                N * H * W, K, Cl, float(min_depth), float(max_depth), dtype_code(dtype),
                                stream_ptr()), "depth_head_bwd")
    return dl


def _head_scale(hb, logits_list, P2, base_fx):
    if P2 is None or base_fx is None:
        return
    assert P2.is_cuda and P2.dtype == torch.float32 and P2.is_contiguous() and P2.shape[1:] == (3, 4)
    assert P2.shape[0] == logits_list[0].shape[0]
    hb.P2, hb.base_fx, hb.nimg = P2.data_ptr(), float(base_fx), P2.shape[0]


def depth_head_fwd_multi(logits_list, bins, K, min_depth, max_depth, P2=None, base_fx=None):
    """[logits [N,H,W,Cl] fp32 per scale] -> [(depth, disp)] in one launch (<= 4 scales, same Cl).
    P2 + base_fx: focal-length depth scaling (depth_encoder.py:36-43)"""
    from .binding import FsHeadBatch
    hb = FsHeadBatch()
    _head_scale(hb, logits_list, P2, base_fx)
    hb.n = len(logits_list)
    Cl = logits_list[0].shape[3]
    outs = []
    for s, lg in enumerate(logits_list):
        N, H, W, c = lg.shape
        assert c == Cl and lg.dtype == torch.float32 and lg.is_contiguous()
        depth = torch.empty(N, 1, H, W, dtype=torch.float32, device=lg.device)
        disp = torch.empty(N, 1, H, W, dtype=torch.float32, device=lg.device)
        hb.logits[s], hb.depth[s], hb.disp[s], hb.M[s] = lg.data_ptr(), depth.data_ptr(), disp.data_ptr(), N * H * W
        outs.append((depth, disp))
    check(lib.fs_depth_head_fwd_multi(C.byref(hb), bins.data_ptr(), K, Cl, float(min_depth), float(max_depth),
                                      stream_ptr()), "depth_head_fwd_multi")
    return outs


def depth_head_bwd_multi(logits_list, bins, d_depths, d_disps, K, min_depth, max_depth, dtype, P2=None, base_fx=None):
    """gradients of all scales' logits in one launch; d_depths / d_disps entries may be None"""
    from .binding import FsHeadBatch
    hb = FsHeadBatch()
    _head_scale(hb, logits_list, P2, base_fx)
    hb.n = len(logits_list)
    Cl = logits_list[0].shape[3]
    outs = []
    for s, lg in enumerate(logits_list):
        N, H, W, c = lg.shape
        assert c == Cl
        dl = torch.empty(N, H, W, Cl, dtype=dtype, device=lg.device)
        hb.logits[s], hb.dlogits[s], hb.M[s] = lg.data_ptr(), dl.data_ptr(), N * H * W
        hb.d_depth[s], hb.d_disp[s] = _p(d_depths[s]), _p(d_disps[s])
        outs.append(dl)
    check(lib.fs_depth_head_bwd_multi(C.byref(hb), bins.data_ptr(), K, Cl, float(min_depth), float(max_depth),
                                      dtype_code(dtype), stream_ptr()), "depth_head_bwd_multi")
    return outs


def pose_tail_fwd(x, nframes, invert, scale=0.01):
    B, h, w, Cx = x.shape
    aa = torch.empty(B, nframes, 1, 3, dtype=torch.float32, device=x.device)
    tr = torch.empty(B, nframes, 1, 3, dtype=torch.float32, device=x.device)
    T = torch.empty(B, 4, 4, dtype=torch.float32, device=x.device)
    check(lib.fs_pose_tail_fwd(x.data_ptr(), aa.data_ptr(), tr.data_ptr(), T.data_ptr(), B, h * w, Cx, nframes,
                               int(invert), float(scale), stream_ptr()), "pose_tail_fwd")
    return aa, tr, T


def pose_tail_bwd(x, dT, nframes, invert, dtype, scale=0.01, out=None):
    B, h, w, Cx = x.shape
    dx = torch.empty(B, h, w, Cx, dtype=dtype, device=x.device) if out is None else out
    assert dx.shape == x.shape and dx.is_contiguous() and dx.dtype == dtype and x.is_contiguous()
    check(lib.fs_pose_tail_bwd(x.data_ptr(), dT.data_ptr(), dx.data_ptr(), B, h * w, Cx, nframes, int(invert),
                               float(scale), dtype_code(dtype), stream_ptr()), "pose_tail_bwd")
    return dx


class PhotometricLoss:
    """Fused loss chain for one batch geometry (B, H, W, scales).  forward() returns the loss
    vector (f64 [2S+1]: loss/s, smooth_loss/s, total); backward() returns d depth_s and dT_f.
    want_pred: also materialise the warped images / overlap masks (self.pred, self.ov) — only needed for logging and
    for the ("original_image", f, s) entries the reference leaves in its output dict."""

    def __init__(self, B, H, W, scales, device, min_depth, max_depth, want_pred=True, overlapped_mask=True):
        self.B, self.H, self.W = B, H, W
        # overlapped_mask=False (multi_dataset / nusc configs): reprojection terms are used wherever they are, with
        # border-clamped samples (monodepth2_decoder.py:110-116, 230-235)
        self.overlapped_mask = bool(overlapped_mask)
        self.scales = list(scales)
        S = self.S = len(self.scales)
        self.device = device
        f32, f64, u8 = torch.float32, torch.float64, torch.uint8
        self.geo = torch.zeros(B, 48, dtype=f32, device=device)
        self.want_pred = bool(want_pred)
        self.pred = self.ov = None
        if self.want_pred:
            self.pred = torch.empty(S, 2, B, 3, H, W, dtype=f32, device=device)
            self.ov = torch.empty(S, 2, B, H, W, dtype=u8, device=device)
        self.ident = torch.empty(B, 2, H, W, dtype=f32, device=device)
        self.sel = torch.empty(S, B, H, W, dtype=u8, device=device)
        # accumulators zeroed every step in one memset: loss_sums[S*B] mask_sum[B] disp_sum[S*B] sm_sums[2*S*B] dot[S*B]
        self.n_acc = S * B + B + S * B + 2 * S * B + S * B
        self.acc = torch.zeros(self.n_acc, dtype=f64, device=device)
        o = 0
        self.loss_sums = self.acc[o:o + S * B]; o += S * B
        self.mask_sum = self.acc[o:o + B]; o += B
        self.disp_sum = self.acc[o:o + S * B]; o += S * B
        self.sm_sums = self.acc[o:o + 2 * S * B]; o += 2 * S * B
        self.dot = self.acc[o:o + S * B]
        self.out = torch.zeros(2 * S + 1, dtype=f64, device=device)
        self.hw = [(H >> s, W >> s) for s in self.scales]
        self.color = [None] * S
        self.bwd_tiles = int(lib.fs_photo_fused_bwd_tiles(H, W))
        self.dP = torch.zeros(self.S * B * self.bwd_tiles, 2, 12, dtype=f32, device=device)   # per-tile partials
        # one flat buffer behind the per-scale depth gradients: a single memset per backward instead of one per scale
        sizes = [B * h * w for (h, w) in self.hw]
        self._dd_flat = torch.zeros(sum(sizes), dtype=f32, device=device)
        self.d_depth, o = [], 0
        for n_el, (h, w) in zip(sizes, self.hw):
            self.d_depth.append(self._dd_flat[o:o + n_el].view(B, 1, h, w))
            o += n_el
        self.d_disp = [torch.empty(B, 1, h, w, dtype=f32, device=device) for (h, w) in self.hw]
        self.dT = [torch.zeros(B, 4, 4, dtype=f32, device=device) for _ in range(2)]
        self.pyr = [None if s == 0 else torch.empty(B, 3, H >> s, W >> s, dtype=f32, device=device)
                    for s in self.scales]
        self._pa = FsPhotoArgs()
        self._sa = FsSmoothArgs()
        self.seed_buf = torch.zeros(1, dtype=torch.int32, device=device)   # device-resident tie-break seed
        self._prefetched = None
        # fisheye (Mei model): persistent per-sample table pointers / parameters / mask plane, refreshed per step by
        # stage_fisheye() so that a captured hipGraph keeps reading the same addresses
        self.fisheye = False
        self.motion_mask = None      # [B,H,W] fp32 or None: precomputed motion mask (monodepth2_decoder.py:243-246)
        self.lut_table = self.mei = self.warp_mask = None
        self._lut_keep = None
        self._clean_acc = self._clean_dd = False
        register_prezero(self, lambda o: o._prezero(), device)

    def stage_fisheye(self, tables, mei_rows):
        """tables: B device tensors [4,H,W] (fs_mei_lut); mei_rows: host float32 [B,8] = k1 k2 xi g1 g2 u0 v0 0.
        Copies the pointer table and the parameters into persistent device buffers on the current stream."""
        B, dev = self.B, self.device
        if self.lut_table is None:
            self.lut_table = torch.zeros(B, dtype=torch.int64, device=dev)
            self.mei = torch.zeros(B, 8, dtype=torch.float32, device=dev)
            self.warp_mask = torch.empty(B, self.H, self.W, dtype=torch.float32, device=dev)
        assert len(tables) == B and all(t.shape == (4, self.H, self.W) and t.is_contiguous() for t in tables)
        ptrs = [t.data_ptr() for t in tables]
        if self._lut_keep is not None and self._lut_keep[1] == ptrs and bool((self._lut_keep[2] == mei_rows).all()):
            self.fisheye = True
            return                                    # same calibrations as the previous step: nothing to upload
        # fresh pinned staging buffers per upload: the caching host allocator keeps them alive until the asynchronous
        # copy has run (a reused buffer could be overwritten by the next step's staging before that)
        self.lut_table.copy_(torch.tensor(ptrs, dtype=torch.int64).pin_memory(), non_blocking=True)
        self.mei.copy_(torch.from_numpy(mei_rows.copy()).pin_memory(), non_blocking=True)
        self._lut_keep = (list(tables), ptrs, mei_rows.copy())
        self.fisheye = True

    def prefetch(self, img0, srcs, patched_mask):
        """The part of the chain that only needs the batch (accumulator reset, colour pyramid, identity
        reprojection + mask sum): may run on another stream beside the networks; forward() then skips it."""
        st = stream_ptr()
        pa = self._pa
        pa.img0 = img0.data_ptr()
        pa.img_src[0], pa.img_src[1] = srcs[0].data_ptr(), srcs[1].data_ptr()
        pa.patched_mask = _p(patched_mask)
        pa.ident, pa.mask_sum = self.ident.data_ptr(), self.mask_sum.data_ptr()
        pa.geo = self.geo.data_ptr()
        pa.B, pa.H, pa.W, pa.S = self.B, self.H, self.W, self.S
        self._input_only(img0, C.byref(pa), st)
        self._prefetched = (img0.data_ptr(), srcs[0].data_ptr(), srcs[1].data_ptr(), _p(patched_mask))

    def _prezero(self):
        """(ops.prezero_all) the accumulators and the depth-gradient maps are about to be zeroed at the step's head"""
        self._clean_acc = self._clean_dd = True
        return [self.acc, self._dd_flat]

    def _input_only(self, img0, pa, st):
        if getattr(self, "_clean_acc", False):
            _join_pack(img0.device)         # zeroed at the step's head on the pack stream: this stream waits for it
        else:
            self.acc.zero_()
        self._clean_acc = False
        for i, s in enumerate(self.scales):
            if s != 0:
                check(lib.fs_color_pyramid(img0.data_ptr(), self.pyr[i].data_ptr(), self.B, self.H, self.W,
                                           self.hw[i][0], self.hw[i][1], st), "color_pyramid")
        check(lib.fs_photo_identity(pa, st), "photo_identity")

    def _fill(self, img0, srcs, patched_mask, depths, disps, noise_seed, gout):
        pa, sa = self._pa, self._sa
        pa.img0 = img0.data_ptr()
        pa.img_src[0], pa.img_src[1] = srcs[0].data_ptr(), srcs[1].data_ptr()
        pa.patched_mask = _p(patched_mask)
        pa.geo = self.geo.data_ptr()
        pa.no_overlap_mask = 0 if self.overlapped_mask else 1
        pa.pred, pa.ov, pa.ident, pa.sel = _p(self.pred), _p(self.ov), self.ident.data_ptr(), self.sel.data_ptr()
        pa.loss_sums, pa.mask_sum = self.loss_sums.data_ptr(), self.mask_sum.data_ptr()
        pa.dP = self.dP.data_ptr()
        pa.gout = _p(gout)
        pa.B, pa.H, pa.W, pa.S = self.B, self.H, self.W, self.S
        pa.noise_seed = -1 if noise_seed is None else int(noise_seed)
        pa.noise_seed_ptr = self.seed_buf.data_ptr() if noise_seed is None else None
        if self.fisheye:
            pa.lut_ptrs, pa.mei, pa.warp_mask = self.lut_table.data_ptr(), self.mei.data_ptr(), self.warp_mask.data_ptr()
        else:
            pa.lut_ptrs = pa.mei = pa.warp_mask = None
        pa.motion_mask = _p(self.motion_mask)
        sa.disp_sum, sa.sm_sums, sa.dot = self.disp_sum.data_ptr(), self.sm_sums.data_ptr(), self.dot.data_ptr()
        sa.gout = _p(gout)
        sa.B, sa.S = self.B, self.S
        for i, (h, w) in enumerate(self.hw):
            pa.depth[i] = depths[i].data_ptr()
            pa.dh[i], pa.dw[i] = h, w
            pa.d_depth[i] = self.d_depth[i].data_ptr()
            sa.disp[i] = disps[i].data_ptr()
            sa.color[i] = (img0 if self.scales[i] == 0 else self.pyr[i]).data_ptr()
            sa.d_disp[i] = self.d_disp[i].data_ptr()
            sa.h[i], sa.w[i], sa.scale_id[i] = h, w, self.scales[i]

    def forward(self, img0, srcs, P2, Ts, patched_mask, depths, disps, noise_seed=-1, motion_mask=None):
        """img0, srcs[2]: NCHW fp32; P2 [B,3,4] fp32; Ts[2]: [B,4,4] fp32; patched_mask f64 [B,H,W] or None;
        depths/disps: per scale [B,1,h,w] fp32 contiguous; motion_mask: fp32 [B,H,W] or None."""
        st = stream_ptr()
        if motion_mask is not None:
            assert motion_mask.dtype == torch.float32 and motion_mask.is_contiguous() and motion_mask.shape == (self.B, self.H, self.W)
        self.motion_mask = motion_mask
        assert img0.is_contiguous() and all(s.is_contiguous() for s in srcs)
        if patched_mask is not None:
            assert patched_mask.dtype == torch.float64 and patched_mask.is_contiguous()
        self._keep = (img0, srcs, patched_mask, depths, disps, noise_seed)
        self._fill(img0, srcs, patched_mask, depths, disps, noise_seed, None)
        pre, self._prefetched = self._prefetched, None
        have_inputs = pre == (img0.data_ptr(), srcs[0].data_ptr(), srcs[1].data_ptr(), _p(patched_mask))
        pa, sa = C.byref(self._pa), C.byref(self._sa)
        # (noise_seed None: the setup kernel bumps the device seed — fresh noise every step, also on a graph replay)
        check(lib.fs_photo_setup(P2.data_ptr(), Ts[0].data_ptr(), Ts[1].data_ptr(), self.geo.data_ptr(), self.B,
                                 self.seed_buf.data_ptr() if noise_seed is None else None, int(self.fisheye), st),
              "photo_setup")
        if self.fisheye:
            check(lib.fs_mei_stage_mask(self.lut_table.data_ptr(), _p(patched_mask), self.warp_mask.data_ptr(), self.B,
                                        self.H, self.W, st), "mei_stage_mask")
        if not have_inputs:
            self._input_only(img0, pa, st)
        # the edge-aware smoothness term needs only the disparities and the colour pyramid: on the (by now idle) pose
        # stream beside the photometric kernels instead of after them — this stretch of the step is serial
        fork = self._fork(img0.device)
        if fork is not None:
            fork[1].wait_stream(fork[0])
            with torch.cuda.stream(fork[1]):
                st2 = stream_ptr()
                check(lib.fs_smooth_mean(sa, st2), "smooth_mean")
                check(lib.fs_smooth_fwd(sa, st2), "smooth_fwd")
        else:
            check(lib.fs_smooth_mean(sa, st), "smooth_mean")
            check(lib.fs_smooth_fwd(sa, st), "smooth_fwd")
        N_px = float(self.B * self.H * self.W)
        # algorithmic bytes: per scale, target 12N + 2 sources 24N + depth 4N/4^s + result 4N
        fwd_bytes = sum(40.0 * N_px + 4.0 * N_px / (4 ** s) for s in self.scales)
        _timed("photo_fused_fwd", fwd_bytes, lambda: check(lib.fs_photo_fused_fwd(pa, st), "photo_fused_fwd"))
        if fork is not None:
            fork[0].wait_stream(fork[1])         # smoothness sums (side stream) before the finalize kernel
        # fresh result tensors per call (the previous step's stay valid for whoever kept them) — the kernel writes
        # the per-scale vector and the scalar the caller differentiates, so nothing has to be cloned on the device
        self.out = torch.empty(2 * self.S + 1, dtype=torch.float64, device=img0.device)
        self.total = torch.empty((), dtype=torch.float64, device=img0.device)
        check(lib.fs_loss_finalize(self.loss_sums.data_ptr(), self.mask_sum.data_ptr(), self.sm_sums.data_ptr(), sa,
                                   self.out.data_ptr(), self.total.data_ptr(), st), "loss_finalize")
        return self.out

    @staticmethod
    def _fork(device):
        """(current stream, pose stream) when the smoothness kernels may run beside the photometric ones"""
        from ..engine.runtime import RT
        if not (RT.overlap and device.type == "cuda"):
            return None
        cur, side = torch.cuda.current_stream(device), RT.side_stream(device)
        return None if cur.cuda_stream == side.cuda_stream else (cur, side)

    def backward(self, gout=None):
        """gout: device f64 scalar (upstream grad of total loss) or None (=1)."""
        st = stream_ptr()
        img0, srcs, patched_mask, depths, disps, noise_seed = self._keep
        self._fill(img0, srcs, patched_mask, depths, disps, noise_seed, gout)
        pa, sa = C.byref(self._pa), C.byref(self._sa)
        if getattr(self, "_clean_dd", False):
            _join_pack(img0.device)
        else:
            self._dd_flat.zero_()
        self._clean_dd = False
        fork = self._fork(img0.device)
        if fork is not None:                      # smoothness backward beside the photometric backward
            fork[1].wait_stream(fork[0])
            with torch.cuda.stream(fork[1]):
                check(lib.fs_smooth_bwd(sa, stream_ptr()), "smooth_bwd")
        N_px = float(self.B * self.H * self.W)
        bwd_bytes = sum(40.0 * N_px + 8.0 * N_px / (4 ** s) for s in self.scales)
        _timed("photo_fused_bwd", bwd_bytes, lambda: check(lib.fs_photo_fused_bwd(pa, st), "photo_fused_bwd"))
        check(lib.fs_photo_pose_grad(self.geo.data_ptr(), self.dP.data_ptr(), self.dT[0].data_ptr(),
                                     self.dT[1].data_ptr(), self.B, self.S, self.bwd_tiles, st), "photo_pose_grad")
        if fork is not None:
            fork[0].wait_stream(fork[1])
        else:
            check(lib.fs_smooth_bwd(sa, st), "smooth_bwd")
        return self.d_depth, self.d_disp, self.dT


def sumsq(g, out, step_counter=None):
    """out += sum(g^2); step_counter: device int32 bumped by one in the same launch"""
    check(lib.fs_sumsq(g.data_ptr(), g.numel(), out.data_ptr(), _p(step_counter), stream_ptr()), "sumsq")


def adam_step(p, g, m, v, lr, b1, b2, eps, wd, step, max_norm=0.0, sumsq_buf=None, grad_scale=1.0, step_buf=None,
              lr_buf=None):
    check(lib.fs_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(b1),
                           float(b2), float(eps), float(wd), int(step), float(max_norm or 0.0), _p(sumsq_buf),
                           float(grad_scale), _p(step_buf), _p(lr_buf), stream_ptr()), "adam_step")


COPY_MAX = 16


def copy_multi(pairs):
    """[(dst, src)] same-shape/dtype contiguous device tensors -> one launch per 16 copies (current stream);
    pairs the kernel cannot take (unaligned views, dtype change, host source) go through Tensor.copy_."""
    fast = []
    for dst, src in pairs:
        if (src.is_cuda and src.dtype == dst.dtype and src.shape == dst.shape and src.is_contiguous()
                and dst.is_contiguous() and src.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0 and src.numel() > 0):
            fast.append((dst, src))
        else:
            dst.copy_(src, non_blocking=True)
    for i in range(0, len(fast), COPY_MAX):
        chunk = fast[i:i + COPY_MAX]
        n = len(chunk)
        srcs = (C.c_void_p * n)(*[s.data_ptr() for _, s in chunk])
        dsts = (C.c_void_p * n)(*[d.data_ptr() for d, _ in chunk])
        nbytes = (C.c_int64 * n)(*[s.numel() * s.element_size() for _, s in chunk])
        check(lib.fs_copy_multi(srcs, dsts, nbytes, n, stream_ptr()), "copy_multi")


def zero_multi(tensors):
    """contiguous 16-byte aligned device tensors -> zeroed, one launch per 16 (current stream)"""
    fast = []
    for t in tensors:
        if t.numel() == 0:
            continue
        if t.is_cuda and t.is_contiguous() and t.data_ptr() % 16 == 0:
            fast.append(t)
        else:
            t.zero_()
    for i in range(0, len(fast), COPY_MAX):
        chunk = fast[i:i + COPY_MAX]
        n = len(chunk)
        dsts = (C.c_void_p * n)(*[t.data_ptr() for t in chunk])
        nbytes = (C.c_int64 * n)(*[t.numel() * t.element_size() for t in chunk])
        check(lib.fs_zero_multi(dsts, nbytes, n, stream_ptr()), "zero_multi")


# ---- per-step scratch zeroed together at the step's head (engine/nets.py pack_everything_async -> prezero_all) ----
_PREZERO = []          # [(weakref to the owner, fn(owner) -> [tensors to zero]; fn marks them clean)]
PREZERO_EPOCH = [0]    # bumped by every prezero_all: owners can tell "zeroed at the head of the running step" from "some time ago"


def _indexed(device):
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return d


def register_prezero(owner, fn, device):
    """fn(owner) -> tensors on `device` to zero; it marks them clean and is only called for that device's steps"""
    import weakref
    _PREZERO.append((weakref.ref(owner), fn, _indexed(device)))


def _join_pack(device):
    from ..engine.nets import join_pack
    join_pack(device)


def prezero_all(device):
    """zero every registered scratch buffer that was used since its last zeroing, in one launch on the current stream; the
    owners' own zero_() calls are then skipped once (their `clean` flags)"""
    todo, alive = [], []
    device = _indexed(device)
    PREZERO_EPOCH[0] += 1
    for ref, fn, dev in _PREZERO:
        o = ref()
        if o is None:
            continue
        alive.append((ref, fn, dev))
        if dev != device:        # (fn has side effects — clean flags, bump pointers: another device's owners are left alone)
            continue
        todo.extend(fn(o))
    _PREZERO[:] = alive
    if todo:
        zero_multi(todo)


def counter_incr(buf):
    """device int32 counter += 1 on the current stream (replayable from a hipGraph)"""
    check(lib.fs_counter_incr(buf.data_ptr(), stream_ptr()), "counter_incr")


# ---------------------------------------------------------------------------------------------
# evaluation
# ---------------------------------------------------------------------------------------------
def resize_linear(src, H, W, invert=False):
    """src: device fp32 [h, w] -> [H, W] with OpenCV's INTER_LINEAR rule (invert: 1 / resize(1 / src))."""
    assert src.is_cuda and src.dtype == torch.float32 and src.dim() == 2
    src = src.contiguous()
    dst = torch.empty(H, W, dtype=torch.float32, device=src.device)
    check(lib.fs_resize_linear(src.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1], H, W, int(invert),
                               stream_ptr()), "resize_linear")
    return dst


def depth_eval(pred, gt):
    """pred: device fp32 [B, h, w]; gt: device fp32 [B, H, W].  Returns f64 [B, 16] on the device:
    ratio, err[7] (median-scaled), abs_err[7], n_valid  (kitti_unsupervised_eval.py:47-80)."""
    assert pred.is_cuda and gt.is_cuda and pred.dtype == gt.dtype == torch.float32
    assert pred.dim() == 3 and gt.dim() == 3 and pred.shape[0] == gt.shape[0]
    pred, gt = pred.contiguous(), gt.contiguous()
    out = torch.empty(pred.shape[0], 16, dtype=torch.float64, device=pred.device)
    scratch = torch.empty(gt.numel() * 2, dtype=torch.float32, device=pred.device)
    check(lib.fs_depth_eval(pred.data_ptr(), gt.data_ptr(), pred.shape[0], pred.shape[1], pred.shape[2], gt.shape[1],
                            gt.shape[2], scratch.data_ptr(), out.data_ptr(), stream_ptr()), "depth_eval")
    return out


# ---------------------------------------------------------------------------------------------
# self-distillation
# ---------------------------------------------------------------------------------------------
def sigmoid_head_fwd(logits):
    """logits: fp32 NHWC [N,H,W,Cp] whose channel 0 is the head output -> u [N,1,H,W] fp32"""
    N, H, W, Cp = logits.shape
    assert logits.dtype == torch.float32 and logits.is_contiguous()
    u = torch.empty(N, 1, H, W, dtype=torch.float32, device=logits.device)
    check(lib.fs_sigmoid_head_fwd(logits.data_ptr(), u.data_ptr(), N * H * W, Cp, stream_ptr()), "sigmoid_head_fwd")
    return u


def sigmoid_head_bwd(u, du, Cp, dtype):
    """-> gradient w.r.t. the head's conv output, NHWC [N,H,W,Cp] in `dtype` (channels 1.. are zero)"""
    N, _, H, W = u.shape
    du = du.contiguous().float()
    dl = torch.empty(N, H, W, Cp, dtype=dtype, device=u.device)
    check(lib.fs_sigmoid_head_bwd(u.data_ptr(), du.data_ptr(), dl.data_ptr(), N * H * W, Cp, dtype_code(dtype),
                                  stream_ptr()), "sigmoid_head_bwd")
    return dl


def distill_fwd(pred, teacher, uncertain=None):
    """mean over all elements of |teacher - pred| (/ uncertain + log(uncertain + 1e-5)) as an f64 device scalar"""
    assert pred.shape == teacher.shape and pred.dtype == teacher.dtype == torch.float32
    pred, teacher = pred.contiguous(), teacher.contiguous()
    if uncertain is not None:
        assert uncertain.shape == pred.shape and uncertain.dtype == torch.float32
        uncertain = uncertain.contiguous()
    acc = torch.zeros((), dtype=torch.float64, device=pred.device)
    check(lib.fs_distill_fwd(pred.data_ptr(), teacher.data_ptr(), _p(uncertain), pred.numel(), acc.data_ptr(),
                             stream_ptr()), "distill_fwd")
    return acc / pred.numel()


def distill_bwd(pred, teacher, uncertain, gout):
    d_pred = torch.empty_like(pred)
    d_unc = torch.empty_like(uncertain) if uncertain is not None else None
    check(lib.fs_distill_bwd(pred.data_ptr(), teacher.data_ptr(), _p(uncertain), pred.numel(), _p(gout), d_pred.data_ptr(),
                             _p(d_unc), stream_ptr()), "distill_bwd")
    return d_pred, d_unc
