"""Host-side plan for one convolution layer on the HIP library: packed MFMA weight operands,
K-walk tables and the three launches (forward, dgrad, wgrad).  Activations are NHWC torch
tensors (any n/h/w strides, channel axis contiguous) used purely as device memory."""
import ctypes as C

import numpy as np
import torch

from .binding import lib, check, stream_ptr, raw_stream, FsConvArgs, FsWgradArgs, FS_DTYPE_BF16, FS_DTYPE_F32


class LaunchProfile:
    """Optional per-launch timing with HIP events recorded on the stream the kernels are launched on
    (torch's current stream).  bench.py enables it for a few extra steps after the timed region."""
    active = False
    records = []   # (kind, flops, start_event, end_event)

    @classmethod
    def begin(cls):
        cls.active, cls.records = True, []

    @classmethod
    def end(cls):
        cls.active = False
        torch.cuda.synchronize()
        out = [(k, f, s.elapsed_time(e) * 1e-3) for (k, f, s, e, _) in cls.records]
        cls.tagged = [(k, f, s.elapsed_time(e) * 1e-3, t) for (k, f, s, e, t) in cls.records]
        cls.records = []
        return out


def _timed(kind, flops, fn, tag=None):
    if not LaunchProfile.active:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn()
    e.record()
    LaunchProfile.records.append((kind, flops, s, e, tag() if callable(tag) else tag))
    return r


class Spec:
    """Arguments of ONE launch, built but not issued: `chain` names the C entry points to try in order (fs_<name>; a
    status of FS_EINVAL from all but the last means "not mine"), `a` is the argument struct, `out` what the caller gets
    back.  run_specs() issues one spec, or two as ONE launch through the fs_<name>2 entry points (include/fsnet_hip.h,
    "Two problems per launch"): the depth encoder's and the stacked pose encoder's layer of the same shape."""
    __slots__ = ("chain", "a", "code", "kind", "work", "tag", "out", "what")

    def __init__(self, chain, a, code, kind, work, tag, out, what):
        self.chain, self.a, self.code, self.kind, self.work, self.tag, self.out, self.what = chain, a, code, kind, work, tag, out, what


# entry points that have a two-problem form
DUAL_ENTRIES = {"conv3x3_halo", "conv3x3_s2d", "conv_igemm", "conv_stem", "conv_wgrad", "bn_apply", "bn_bwd_reduce", "bn_bwd_apply"}


def _run_chain(chain, a0, a1, code, what):
    """-> name of the entry point that took the launch"""
    for k, name in enumerate(chain):
        if a1 is None:
            status = getattr(lib, "fs_" + name)(C.byref(a0), code, stream_ptr())
        else:
            status = getattr(lib, "fs_" + name + "2")(C.byref(a0), C.byref(a1), code, stream_ptr())
        if status == 1 and k + 1 < len(chain):      # FS_EINVAL = "not mine": the next entry point of the chain
            continue
        check(status, what)
        return name
    raise AssertionError("empty launch chain")


def run_specs(specs):
    """issue the launches; two specs whose chains agree run as one launch (the library falls back to two when the
    problems do not agree on a kernel instantiation — results never depend on the pairing)"""
    specs = [sp for sp in specs if sp is not None]
    if len(specs) == 2 and specs[0].chain == specs[1].chain and specs[0].code == specs[1].code \
            and all(n in DUAL_ENTRIES for n in specs[0].chain):
        s0, s1 = specs
        if not LaunchProfile.active:
            _run_chain(s0.chain, s0.a, s1.a, s0.code, s0.what)
            return
        taken = [None]

        def go():
            taken[0] = _run_chain(s0.chain, s0.a, s1.a, s0.code, s0.what)
        kind = s0.kind(s1) if callable(s0.kind) else s0.kind
        _timed(kind, s0.work + s1.work, go, tag=lambda: "%s || %s" % (s0.tag() if callable(s0.tag) else s0.tag,
                                                                      s1.tag() if callable(s1.tag) else s1.tag))
        return
    for sp in specs:
        if not LaunchProfile.active:
            _run_chain(sp.chain, sp.a, None, sp.code, sp.what)
            continue
        n0 = len(LaunchProfile.records)
        taken = [None]

        def go(sp=sp):
            taken[0] = _run_chain(sp.chain, sp.a, None, sp.code, sp.what)
        kind = sp.kind(None) if callable(sp.kind) else sp.kind
        _timed(kind, sp.work, go, tag=sp.tag)
        if len(sp.chain) > 1 and taken[0] != sp.chain[0] and LaunchProfile.records[n0:]:
            # the first entry point declined: the record belongs to the one that ran
            r = LaunchProfile.records[-1]
            LaunchProfile.records[-1] = (taken[0],) + tuple(r[1:])


def dtype_code(dtype):
    if dtype == torch.bfloat16:
        return FS_DTYPE_BF16
    if dtype == torch.float32:
        return FS_DTYPE_F32
    raise ValueError("unsupported compute dtype %s (bf16 or fp32)" % dtype)


def roundup(a, b):
    return (a + b - 1) // b * b


def make_ktab(R, S, cs_p, eg, ngroups):
    """One int per 16-byte K group: c | r<<16 | s<<24, -1 past the last tap."""
    tab = np.full(ngroups, -1, dtype=np.int32)
    real = R * S * cs_p // eg
    k0 = np.arange(min(real, ngroups), dtype=np.int64) * eg
    tap = k0 // cs_p
    c = k0 % cs_p
    r = tap // S
    s = tap % S
    tab[: len(k0)] = (c | (r << 16) | (s << 24)).astype(np.int32)
    return tab


def make_unit_table(R, S, cs_p, eg, ug, nunits, sgn, sH, sW, elem_bytes):
    """int2 per K unit (ug consecutive 16-byte groups of one tap): byte delta of the tap + channel, and the
    signed (r, s) pair.  INT32_MIN marks K padding."""
    tab = np.zeros((nunits, 2), dtype=np.int32)
    tab[:, 0] = np.iinfo(np.int32).min
    real = R * S * cs_p // (eg * ug)
    k0 = np.arange(min(real, nunits), dtype=np.int64) * eg * ug
    tap, c = k0 // cs_p, k0 % cs_p
    r, s = sgn * (tap // S), sgn * (tap % S)
    tab[: len(k0), 0] = ((r * sH + s * sW + c) * elem_bytes).astype(np.int32)
    tab[: len(k0), 1] = ((r & 0xffff) | (s << 16)).astype(np.int64).astype(np.int32)
    return tab


def _span_bytes(t):
    n, h, w, c = t.shape
    return ((n - 1) * t.stride(0) + (h - 1) * t.stride(1) + (w - 1) * t.stride(2) + c) * t.element_size()


def _nhwc_strides(t):
    assert t.dim() == 4 and t.stride(3) == 1, "activation must be NHWC with contiguous channels"
    return t.stride(0), t.stride(1), t.stride(2)


import os

BN_EPS = 1e-5            # nn.BatchNorm2d defaults (resnet.py:17-19, blocks.py:46)
BN_MOMENTUM = 0.1

PRO_MAX_CI = 512      # conv_pro.h pro_args_ok: the operand prologue's coefficient table (2 floats per source channel in LDS)
# Tests only: FsConvArgs.force_impl of every 3x3 / stride-1 launch (include/fsnet_hip.h: 1 = 16x16-tile kernel, 2-4 = 32x32-tile
# kernel in tile configuration 1-3, 5 = persistent one-chunk kernel).  0 in the product: the entry point chooses.
FORCE_3X3 = 0
_WGRAD_WS = {}


def wgrad_workspace(device, elems=1 << 23):
    """per-device fp32 scratch for split-K partial slabs (32 MB; stream-ordered reuse)"""
    key = (device, raw_stream(device.index))   # one scratch per stream: no cross-stream reuse
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < elems:
        ws = _WGRAD_WS[key] = torch.empty(elems, dtype=torch.float32, device=device)
    return ws


class ConvOp:
    def __init__(self, Ci, Co, R, S, stride, pad, dtype, device, need_dgrad=True):
        assert stride in (1, 2)
        self.Ci, self.Co, self.R, self.S, self.stride, self.pad = Ci, Co, R, S, stride, pad
        self.dtype = dtype
        self.code = dtype_code(dtype)
        self.device = device
        self.EG = 8 if dtype == torch.bfloat16 else 4
        eg = self.EG
        self.Ci_p = roundup(Ci, eg)      # channels of the input activation buffer
        self.Co_p = roundup(Co, 16)      # channels of the output activation buffer
        # forward operand.  Stage depth kg (16-byte K groups per LDS stage): 8 (= 64 bf16 channels) when the
        # K walk is long enough to profit and the channel tile allows it, else 4.
        def pick_kg(ktot, rows_p, cs):
            return 8 if (ktot >= 64 * eg and rows_p % 32 == 0 and cs % (4 * eg) == 0) else 4
        self.kg_f = pick_kg(R * S * self.Ci_p, self.Co_p, self.Ci_p)
        chunk = self.kg_f * eg
        self.nch_f = (R * S * self.Ci_p + chunk - 1) // chunk
        self.kf_p = self.nch_f * chunk
        self.w_f = torch.zeros(self.Co_p, self.kf_p, dtype=dtype, device=device)
        # dgrad operand: rows = ci, K = (r, s, co)
        self.need_dgrad = need_dgrad
        if need_dgrad:
            self.rows_d = roundup(self.Ci_p, 16)
            self.kg_d = pick_kg(R * S * self.Co_p, self.rows_d, self.Co_p)
            chunk = self.kg_d * eg
            self.nch_d = (R * S * self.Co_p + chunk - 1) // chunk
            self.kd_p = self.nch_d * chunk
            self.w_d = torch.zeros(self.rows_d, self.kd_p, dtype=dtype, device=device)
        self._tabs = {}
        eb = 2 if dtype == torch.bfloat16 else 4
        # stride-2 3x3 dgrad by output parity: dx[2y'+py, 2x'+px] only receives the taps with r = py+1 (mod 2),
        # s = px+1 (mod 2) -> four stride-1 problems with 1 / 2 / 2 / 4 taps whose weights are K slices of one
        # operand packed in class order (layout.hip tap_at).  The parity-test formulation multiplies all 9 taps for
        # every pixel and masks 3/4 of them.
        self.s2_classes = bool(need_dgrad and stride == 2 and R == 3 and S == 3 and pad == 1
                               and (self.Co_p * eb) % (self.kg_d * 16) == 0)
        # ... and the 1x1 / stride-2 downsample projections (resnet.py:152-160): dx is dy Wt on the (even, even) lattice and
        # zero (+ addend) elsewhere — class (0,0) with its one tap, three classes with none.  The parity-test formulation
        # multiplies four times the rows (ResNet-50 @320x1024: 170-270 us per launch at 80-130 TFLOP/s of mostly zeros).
        self.s2_classes_1x1 = bool(need_dgrad and stride == 2 and R == 1 and S == 1 and pad == 0
                                   and (self.Co_p * eb) % (self.kg_d * 16) == 0)
        # ... all four classes from one staged dY halo (conv3x3_s2d.hip) when dY has whole 64-byte channel chunks
        self.s2d = bool(self.s2_classes and (self.Co_p * eb) % 64 == 0 and self.rows_d % 32 == 0
                        and self.Ci_p % 8 == 0)
        halo_ok = (R == 3 and S == 3 and stride == 1)
        def chunks_ok(c):      # whole 64-byte chunks, or exactly half of one (16 bf16 channels)
            return (c * eb) % 64 == 0 or c * eb == 32
        # 7x7/s2 stem over 16-byte pixels: LDS-resident weights + im2col from an LDS patch (conv_stem.hip)
        self.stem_lds = (R == 7 and S == 7 and stride == 2 and pad == 3 and dtype == torch.bfloat16 and self.Ci_p == 8
                         and Co == 64)
        self.halo_f = halo_ok and chunks_ok(self.Ci_p) and self.Co_p % 16 == 0
        # stride-2 3x3 forward (ResNet stage entries) on the LDS-halo kernel too: whole 64-byte channel chunks, pad 1
        self.halo_f_s2 = (R == 3 and S == 3 and stride == 2 and pad == 1 and (self.Ci_p * eb) % 64 == 0
                          and self.Co_p % 16 == 0)
        self.halo_d = halo_ok and need_dgrad and chunks_ok(self.Co_p) and roundup(self.Ci_p, 16) % 16 == 0
        # wgrad columns (r, s, ci): same grouping as the forward K walk without chunk padding
        self.ncolgroups = R * S * self.Ci_p // eg
        self.ktab_w = torch.from_numpy(make_ktab(R, S, self.Ci_p, eg, self.ncolgroups)).to(device)

    def _table(self, mode, sH, sW):
        """K-unit table for (forward | dgrad) over a source with row/pixel strides (sH, sW) [elements]."""
        key = (mode, sH, sW)
        t = self._tabs.get(key)
        if t is None:
            eb = 2 if self.dtype == torch.bfloat16 else 4
            if mode == "f":
                ug = 4 if self.kg_f == 8 else 1
                tab = make_unit_table(self.R, self.S, self.Ci_p, self.EG, ug, self.nch_f * self.kg_f // ug, 1, sH, sW, eb)
            else:
                ug = 4 if self.kg_d == 8 else 1
                sh = 1 if self.stride == 2 else 0
                tab = make_unit_table(self.R, self.S, self.Co_p, self.EG, ug, self.nch_d * self.kg_d // ug, -1,
                                      sH >> sh, sW >> sh, eb)
            t = self._tabs[key] = torch.from_numpy(tab).to(self.device)
        return t

    # ------------------------------------------------------------------
    def pack(self, weight):
        """weight: fp32 OIHW master parameter on device."""
        assert weight.dtype == torch.float32 and weight.is_contiguous()
        st = stream_ptr()
        check(lib.fs_pack_weights(weight.data_ptr(), self.w_f.data_ptr(), self.Co, self.Ci, self.R, self.S,
                                  self.Co_p, self.Ci_p, self.kf_p, 0, self.code, st), "pack_weights")
        if self.need_dgrad:
            check(lib.fs_pack_weights(weight.data_ptr(), self.w_d.data_ptr(), self.Co, self.Ci, self.R, self.S,
                                      self.rows_d, self.Co_p, self.kd_p, 2 if self.s2_classes else 1, self.code, st),
                  "pack_weights_t")

    def out_hw(self, H, W):
        Ho = (H + 2 * self.pad - self.R) // self.stride + 1
        Wo = (W + 2 * self.pad - self.S) // self.stride + 1
        return Ho, Wo

    # ------------------------------------------------------------------
    def forward(self, x, **kw):
        """one convolution forward, issued now (see forward_spec for the arguments)"""
        sp = self.forward_spec(x, **kw)
        run_specs([sp])
        return sp.out

    def forward_spec(self, x, out=None, bias=None, addend=None, stats=None, relu=False, out_f32=False, stat_groups=1, pro=None):
        """stat_groups G > 1: the batch is G stacked BatchNorm invocations; stats is [G][SLOTS][2][Co].
        pro = (BnState with scale / shift, relu): x is the RAW output of the convolution in front of a BatchNorm (+ ReLU)
        that is applied while the operand is staged (fs_conv3x3_halo prologue; can_fold_input())."""
        N, H, W, Cs = x.shape
        assert Cs == self.Ci_p and x.dtype == self.dtype, (x.shape, self.Ci_p, x.dtype)
        Ho, Wo = self.out_hw(H, W)
        if out is None:
            out = torch.empty(N, Ho, Wo, self.Co_p, dtype=torch.float32 if out_f32 else self.dtype,
                              device=x.device)
        halo = self.halo_f or self.halo_f_s2
        stem = (self.stem_lds and bias is None and addend is None and not relu and not out_f32
                and out.shape[3] == 64 and out.dtype == self.dtype)
        group_rows, grp_imgs = 0, 0
        if stats is not None and stat_groups > 1:
            assert N % stat_groups == 0 and addend is None
            n = N // stat_groups
            group_rows = n * Ho * Wo
            if not halo and not stem and group_rows % 256 != 0:
                grp_imgs = n       # a pixel tile could straddle two groups: one group per blockIdx.z instead
        a = FsConvArgs()
        a.src, a.wgt, a.dst = x.data_ptr(), self.w_f.data_ptr(), out.data_ptr()
        a.bias = bias.data_ptr() if bias is not None else None
        a.addend = addend.data_ptr() if addend is not None else None
        a.stats = stats.data_ptr() if stats is not None else None
        a.sN, a.sH, a.sW = _nhwc_strides(x)
        a.ktab = self._table("f", a.sH, a.sW).data_ptr()
        a.src_bytes, a.wgt_bytes = _span_bytes(x), self.w_f.numel() * self.w_f.element_size()
        a.dN, a.dH, a.dW = _nhwc_strides(out)
        if addend is not None:
            a.aN, a.aH, a.aW = _nhwc_strides(addend)
        a.Hs, a.Ws, a.Hd, a.Wd = H, W, Ho, Wo
        a.M = N * Ho * Wo
        a.Co, a.Co_p, a.nchunks, a.kg = out.shape[3], self.Co_p, self.nch_f, self.kg_f
        a.hb_mul, a.hb_add, a.sgn, a.dshift = self.stride, -self.pad, 1, 0
        a.relu, a.out_f32 = int(relu), int(out_f32)
        a.N, a.Cs = N, self.Ci_p
        a.stat_group_rows = group_rows
        if pro is not None:
            pst, prelu = pro[0], pro[1]
            assert halo and pst.scale is not None and N % pst.groups == 0
            a.pro_mode, a.pro_relu = 1, int(prelu)
            a.pro_group_imgs = N // pst.groups if pst.groups > 1 else 0
            if len(pro) > 2:
                # the consumer finalises the statistics itself (no fs_bn_finalize launch): every block derives scale /
                # shift from the f64 sums, block 0 saves them with mean / invstd for the backward and updates the
                # running statistics
                pstats, pbn, pcount, ptrack = pro[2:]
                a.pro_stats = pstats.data_ptr()
                a.pro_gamma, a.pro_beta = pbn["weight"].data_ptr(), pbn["bias"].data_ptr()
                a.pro_mean, a.pro_invstd = pst.mean.data_ptr(), pst.invstd.data_ptr()
                a.pro_save_a, a.pro_save_b = pst.scale.data_ptr(), pst.shift.data_ptr()
                if ptrack:
                    a.pro_running_mean, a.pro_running_var = pbn["running_mean"].data_ptr(), pbn["running_var"].data_ptr()
                    a.pro_nbt = pbn["num_batches_tracked"].data_ptr()
                pst.count = float(pcount)
                a.pro_count, a.pro_eps, a.pro_momentum = float(pcount), BN_EPS, BN_MOMENTUM
            else:
                a.pro_a, a.pro_b = pst.scale.data_ptr(), pst.shift.data_ptr()
        flops = 2.0 * a.M * self.Co * self.R * self.S * self.Ci
        if grp_imgs:
            a.stat_group_rows, a.grp_imgs, a.M = 0, grp_imgs, grp_imgs * Ho * Wo
        tag = lambda: "fwd  %s x[%d,%d,%d,%d]" % (self.describe(), N, x.shape[1], x.shape[2], x.shape[3])
        if self._wants_1x1(a):
            return Spec(["conv1x1", "conv_igemm"], a, self.code, "conv1x1", flops, tag, out, "conv_fwd")
        if stem:
            return Spec(["conv_stem"], a, self.code, "conv_stem", flops, tag, out, "conv_stem")
        if halo:
            a.force_impl = FORCE_3X3
            return Spec(["conv3x3_halo"], a, self.code, lambda other: self._kind3x3(a, other), flops, tag, out, "conv_fwd")
        return Spec(["conv_igemm"], a, self.code, "conv_igemm", flops, tag, out, "conv_fwd")

    def _kind3x3(self, a, other=None):
        """launch-profile kind of a 3x3 / stride-1 launch: which of the three kernels fs_conv3x3_halo runs it on
        (other: the Spec sharing the launch)"""
        if not LaunchProfile.active:
            return "conv3x3_halo"
        if a.hb_mul == 2:
            return "conv3x3_s2"          # stage-entry stride-2 forward: not part of the stride-1 family's roofline figure
        plan = (C.c_int32 * 4)()
        if other is not None and lib.fs_conv3x3_halo2_plan(C.byref(a), C.byref(other.a), self.code, plan) == 0:
            return {1: "conv3x3_t32", 2: "conv3x3_p1"}.get(int(plan[0]), "conv3x3_halo")
        check(lib.fs_conv3x3_halo_plan(C.byref(a), self.code, plan), "conv3x3_plan")
        return {1: "conv3x3_t32", 2: "conv3x3_p1"}.get(int(plan[0]), "conv3x3_halo")

    def plan_3x3(self, N, H, W, forward=True, pro_mode=0, Co_out=None):
        """the launch fs_conv3x3_halo makes for a forward (x [N,H,W,Ci_p]) or data-gradient (dy [N,H,W,Co_p]) call of this
        3x3 / stride-1 layer on dense tensors: {"kernel": "t32" | "halo", "blocks", "pix", "co"} — nothing is launched"""
        assert self.R == 3 and self.S == 3 and (self.stride == 1 or forward)
        eb = 2 if self.dtype == torch.bfloat16 else 4
        a = FsConvArgs()
        a.src, a.wgt, a.dst = 16, 16, 16             # only tested against NULL
        Cs, rows = (self.Ci_p, self.Co_p) if forward else (self.Co_p, self.rows_d)
        Ho, Wo = self.out_hw(H, W) if forward else (H, W)
        a.sN, a.sH, a.sW = H * W * Cs, W * Cs, Cs
        a.dN, a.dH, a.dW = Ho * Wo * rows, Wo * rows, rows
        a.src_bytes = N * H * W * Cs * eb
        a.wgt_bytes = (self.w_f if forward else self.w_d).numel() * eb
        a.Hs, a.Ws, a.Hd, a.Wd, a.M = H, W, Ho, Wo, N * Ho * Wo
        a.Co = Co_out if Co_out is not None else (self.Co_p if forward else self.Ci_p)
        a.Co_p, a.nchunks, a.kg = rows, (self.nch_f if forward else self.nch_d), (self.kg_f if forward else self.kg_d)
        a.hb_mul, a.hb_add, a.sgn = (self.stride if forward else 1), (-self.pad if forward else self.pad), (1 if forward else -1)
        a.N, a.Cs = N, Cs
        if pro_mode:
            a.pro_mode, a.pro_a, a.pro_b = pro_mode, 16, 16
        a.force_impl = FORCE_3X3
        plan = (C.c_int32 * 4)()
        check(lib.fs_conv3x3_halo_plan(C.byref(a), self.code, plan), "conv3x3_plan")
        return {"kernel": {0: "halo", 1: "t32", 2: "p1"}[int(plan[0])], "blocks": int(plan[1]), "pix": int(plan[2]), "co": int(plan[3])}

    def _wants_1x1(self, a):
        """1x1 convolutions (forward, stride-1 data gradient) on the row-streaming GEMM kernel (conv1x1.hip); the launch
        chain is then fs_conv1x1 -> fs_conv_igemm (the kernel may still say FS_EINVAL = "not mine")"""
        if not (self.R == 1 and self.S == 1 and self.pad == 0 and self.dtype == torch.bfloat16):
            return False
        if a.stat_group_rows and a.stat_group_rows % 128 != 0:
            return False
        if a.dshift != 0 or a.ncls > 1:
            return False              # (a stride-2 data gradient: the kernel would decline it — straight to the implicit GEMM)
        return True                   # LDS-staged GEMM (or, for what it declines, the row-streaming kernel): any K extent

    def can_fold_input(self, N, H, W):
        """whether this convolution can take its input as (raw convolution output, BatchNorm statistics, ReLU) — forward
        AND weight gradient apply the normalisation while staging the operand (both 3x3 LDS-halo forward kernels and the
        LDS-halo weight-gradient kernel: bf16, 3x3 / stride 1, whole 64-byte channel chunks, >= 64 output channels),
        and the data gradient derives the ReLU mask in its epilogue"""
        eb = 2 if self.dtype == torch.bfloat16 else 4
        return (self.dtype == torch.bfloat16 and self.R == 3 and self.S == 3 and self.stride == 1
                and self.Ci == self.Ci_p and (self.Ci_p * eb) % 64 == 0 and self.Ci_p <= PRO_MAX_CI
                and self.Co_p % 64 == 0 and self.Co % 4 == 0 and self.need_dgrad and N * H * W * max(self.Co_p, self.Ci_p) * eb < 0x7fffffff)

    def can_fuse_bn_bwd(self, N, H, W, groups):
        """whether dgrad(..., bn_fuse=) may carry the BatchNorm-backward sums of a [N,H,W,Ci_p] gradient"""
        if self.Ci_p != self.Ci:
            return False
        rows = (N // groups) * H * W
        if (self.s2_classes or self.s2_classes_1x1) and H % 2 == 0 and W % 2 == 0:
            rows //= 4                                  # one launch per output-parity class
        if groups > 1 and not self.halo_d and not self.s2d and rows % 256 != 0:
            return False      # an implicit-GEMM tile could straddle two statistics groups
        return True

    # (py, px) -> [(position in the class-ordered operand, dy row offset, dy column offset)]
    _S2_CLASSES = (((0, 0), ((0, 0, 0),)),
                   ((0, 1), ((1, 0, 1), (2, 0, 0))),
                   ((1, 0), ((3, 1, 0), (4, 0, 0))),
                   ((1, 1), ((5, 1, 1), (6, 1, 0), (7, 0, 1), (8, 0, 0))))

    _S2_CLASSES_1X1 = (((0, 0), ((0, 0, 0),)), ((0, 1), ()), ((1, 0), ()), ((1, 1), ()))

    def _classes(self):
        return self._S2_CLASSES if self.R == 3 else self._S2_CLASSES_1X1

    def _class_tables(self, sH, sW):
        """unit tables of the four parity classes, concatenated; returns (table, [offset in units], [K stages])"""
        key = ("c", sH, sW)
        t = self._tabs.get(key)
        if t is None:
            eb = 2 if self.dtype == torch.bfloat16 else 4
            ug = 4 if self.kg_d == 8 else 1
            stage = self.kg_d * 16
            tabs, offs, nchs = [], [], []
            pos = 0
            for _, taps in self._classes():
                nunits = len(taps) * self.Co_p // (self.EG * ug)
                if nunits == 0:         # a class without taps: its blocks only run the epilogue (zeros + addend)
                    tabs.append(np.zeros((0, 2), dtype=np.int32)); offs.append(pos); nchs.append(0)
                    continue
                tab = np.zeros((nunits, 2), dtype=np.int32)
                k0 = np.arange(nunits, dtype=np.int64) * self.EG * ug
                tl, c = k0 // self.Co_p, k0 % self.Co_p
                dr = np.array([tp[1] for tp in taps], dtype=np.int64)[tl]
                ds = np.array([tp[2] for tp in taps], dtype=np.int64)[tl]
                tab[:, 0] = ((dr * sH + ds * sW + c) * eb).astype(np.int32)
                tab[:, 1] = ((dr & 0xffff) | (ds << 16)).astype(np.int32)
                tabs.append(tab); offs.append(pos); nchs.append(len(taps) * self.Co_p * eb // stage)
                pos += nunits
            cat = np.concatenate(tabs, 0)
            cat = np.concatenate([cat, np.zeros((max(1, 16 - len(cat)), 2), dtype=np.int32)], 0)    # (never an empty table)
            t = self._tabs[key] = (torch.from_numpy(cat).to(self.device), offs, nchs)
        return t

    def can_fold_ds_dgrad(self, ds_op, dy, dc_ds):
        """whether this 3x3 / stride-2 convolution's data gradient can carry the data gradient of the block's 1x1 /
        stride-2 downsample projection ds_op (its dY: dc_ds) in the same launch (fs_conv3x3_s2d with FsConvArgs.ds_src)"""
        return (self.s2d and ds_op.need_dgrad and ds_op.R == 1 and ds_op.S == 1 and ds_op.stride == 2 and ds_op.pad == 0
                and ds_op.dtype == self.dtype and ds_op.Ci_p == self.Ci_p and ds_op.Co_p == self.Co_p
                and ds_op.rows_d == self.rows_d and dy.is_contiguous() and dc_ds.is_contiguous()
                and dc_ds.shape == dy.shape and dc_ds.dtype == dy.dtype)

    def _dgrad_s2_classes(self, dy, H, W, out, addend, mask, bn_fuse, ds=None):
        """all four parity classes in ONE launch (blockIdx.y = class) -> Spec.  ds = (downsample ConvOp, its dY): see
        can_fold_ds_dgrad()"""
        N, Ho, Wo, Cd = dy.shape
        eb = dy.element_size()
        tab, offs, nchs = self._class_tables(*_nhwc_strides(dy)[1:])
        o = out[:, 0::2, 0::2]                       # sub-lattice of class (0,0); the kernel shifts it per class
        a = FsConvArgs()
        a.src, a.dst, a.wgt = dy.data_ptr(), o.data_ptr(), self.w_d.data_ptr()
        a.wgt_row_bytes = self.kd_p * eb
        a.wgt_bytes = (self.rows_d - 1) * self.kd_p * eb + (4 if self.R == 3 else 1) * self.Co_p * eb      # longest class slice
        a.sN, a.sH, a.sW = _nhwc_strides(dy)
        a.ktab = tab.data_ptr()
        a.src_bytes = _span_bytes(dy)
        a.dN, a.dH, a.dW = _nhwc_strides(o)
        if addend is not None:
            ad = addend[:, 0::2, 0::2]
            a.addend = ad.data_ptr()
            a.aN, a.aH, a.aW = _nhwc_strides(ad)
        if mask is not None:
            mk = mask[:, 0::2, 0::2]
            a.mask = mk.data_ptr()
            a.mN, a.mH, a.mW = _nhwc_strides(mk)
        a.Hs, a.Ws, a.Hd, a.Wd = Ho, Wo, H // 2, W // 2
        a.M = N * (H // 2) * (W // 2)
        a.Co, a.Co_p, a.nchunks, a.kg = out.shape[3], self.rows_d, max(nchs), self.kg_d
        a.hb_mul, a.hb_add, a.sgn, a.dshift = 1, 0, 1, 0
        a.N, a.Cs = N, self.Co_p
        a.ncls = 4
        for k, ((py, px), taps) in enumerate(self._classes()):
            assert k == 2 * py + px
            a.cls_nch[k], a.cls_ktab_off[k] = nchs[k], min(offs[k], 0 if not taps else offs[k])
            a.cls_wgt_off[k] = taps[0][0] * self.Co_p * eb if taps else 0
        if bn_fuse is not None:
            c, st, sums = bn_fuse
            a.bnb_x = c.data_ptr()                   # same layout as out: addressed through the dst offsets
            a.bnb_mean, a.bnb_invstd = st.mean.data_ptr(), st.invstd.data_ptr()
            a.stats = sums.data_ptr()
            a.stat_group_rows = (N // st.groups) * (H // 2) * (W // 2) if st.groups > 1 else 0
        flops = 2.0 * N * Ho * Wo * self.Co * self.R * self.S * self.Ci
        if ds is not None:
            ds_op, dc_ds = ds
            assert self.can_fold_ds_dgrad(ds_op, dy, dc_ds) and out.is_contiguous()
            assert (addend is None or addend.is_contiguous()) and (mask is None or mask.is_contiguous())
            a.ds_src, a.ds_wgt, a.ds_wgt_row_bytes = dc_ds.data_ptr(), ds_op.w_d.data_ptr(), ds_op.kd_p * eb
            flops += 2.0 * N * Ho * Wo * self.Co * self.Ci
            # (only fs_conv3x3_s2d carries the projection: a declined launch fails loudly)
            return Spec(["conv3x3_s2d"], a, self.code, "conv3x3_s2d", flops,
                        lambda: "dgrd %s + 1x1/s2 dy[%d,%d,%d,%d]" % (self.describe(), N, Ho, Wo, Cd), out, "conv_dgrad_s2")
        # fs_conv3x3_s2d: all four classes from one staged dY halo (whole 64-byte chunks of dY); the implicit GEMM's class
        # launch takes what it declines
        return Spec(["conv3x3_s2d", "conv_igemm"] if self.s2d else ["conv_igemm"], a, self.code,
                    "conv3x3_s2d" if self.s2d else "conv_igemm", flops,
                    lambda: "dgrd %s dy[%d,%d,%d,%d]" % (self.describe(), N, Ho, Wo, Cd), out, "conv_dgrad_s2")

    def dgrad(self, dy, H, W, **kw):
        """one data gradient, issued now (see dgrad_spec)"""
        sp = self.dgrad_spec(dy, H, W, **kw)
        run_specs([sp])
        return sp.out

    def dgrad_spec(self, dy, H, W, out=None, addend=None, mask=None, bn_fuse=None, mask_bn=False, ds=None):
        """dy: [N,Ho,Wo,Co_p] -> dx [N,H,W,Ci_p] (out may be a strided view; addend is summed in).
        bn_fuse = (c, BnState, sums): dx is the gradient w.r.t. relu(BN(c)) (+ residual): the epilogue also
        accumulates the BatchNorm-backward sums (sum g, sum g*xhat) of that BatchNorm into `sums` (zeroed f64
        [groups][SLOTS][2][C]), i.e. fs_bn_bwd_reduce without a launch and without re-reading g and y."""
        N, Ho, Wo, Cd = dy.shape
        assert Cd == self.Co_p and dy.dtype == self.dtype and self.need_dgrad
        if out is None:
            out = torch.empty(N, H, W, self.Ci_p, dtype=self.dtype, device=dy.device)
        if bn_fuse is not None:
            c, st, _ = bn_fuse
            assert c.is_contiguous() and out.is_contiguous() and c.shape == out.shape and c.dtype == self.dtype
            assert self.can_fuse_bn_bwd(N, H, W, st.groups)
        if self.s2_classes:
            if H % 2 or W % 2 or H != 2 * Ho or W != 2 * Wo:
                raise NotImplementedError("stride-2 3x3 data gradient expects an even input size (%dx%d)" % (H, W))
            return self._dgrad_s2_classes(dy, H, W, out, addend, mask, bn_fuse, ds)
        if self.s2_classes_1x1 and ds is None and H % 2 == 0 and W % 2 == 0 and H == 2 * Ho and W == 2 * Wo and not mask_bn:
            return self._dgrad_s2_classes(dy, H, W, out, addend, mask, bn_fuse, None)
        assert ds is None
        a = FsConvArgs()
        a.src, a.wgt, a.dst = dy.data_ptr(), self.w_d.data_ptr(), out.data_ptr()
        a.bias, a.stats = None, None
        a.addend = addend.data_ptr() if addend is not None else None
        a.sN, a.sH, a.sW = _nhwc_strides(dy)
        a.ktab = self._table("d", a.sH, a.sW).data_ptr()
        a.src_bytes, a.wgt_bytes = _span_bytes(dy), self.w_d.numel() * self.w_d.element_size()
        a.dN, a.dH, a.dW = _nhwc_strides(out)
        if addend is not None:
            a.aN, a.aH, a.aW = _nhwc_strides(addend)
        if mask is not None:
            a.mask = mask.data_ptr()
            a.mN, a.mH, a.mW = _nhwc_strides(mask)
        a.Hs, a.Ws, a.Hd, a.Wd = Ho, Wo, H, W
        a.M = N * H * W
        a.Co, a.Co_p, a.nchunks, a.kg = out.shape[3], self.rows_d, self.nch_d, self.kg_d
        a.hb_mul, a.hb_add, a.sgn, a.dshift = 1, self.pad, -1, (1 if self.stride == 2 else 0)
        a.relu, a.out_f32 = 0, 0
        a.N, a.Cs = N, self.Co_p
        if bn_fuse is not None:
            c, st, sums = bn_fuse
            a.bnb_x, a.bnb_mean, a.bnb_invstd = c.data_ptr(), st.mean.data_ptr(), st.invstd.data_ptr()
            a.stats = sums.data_ptr()
            a.stat_group_rows = (N // st.groups) * H * W if st.groups > 1 else 0
            if mask_bn:
                # the ReLU mask of a folded BatchNorm: sign of scale * c + shift, evaluated in the epilogue from the
                # tensor it reads anyway (the normalised activation was never stored)
                assert mask is None and st.scale is not None and self.halo_d
                a.bnb_scale, a.bnb_shift = st.scale.data_ptr(), st.shift.data_ptr()
        else:
            assert not mask_bn
        flops = 2.0 * N * Ho * Wo * self.Co * self.R * self.S * self.Ci
        halo = self.halo_d
        tag = lambda: "dgrd %s dy[%d,%d,%d,%d]" % (self.describe(), N, Ho, Wo, Cd)
        if self._wants_1x1(a):
            return Spec(["conv1x1", "conv_igemm"], a, self.code, "conv1x1", flops, tag, out, "conv_dgrad")
        if halo:
            a.force_impl = FORCE_3X3
            return Spec(["conv3x3_halo"], a, self.code, lambda other: self._kind3x3(a, other), flops, tag, out, "conv_dgrad")
        return Spec(["conv_igemm"], a, self.code, "conv_igemm", flops, tag, out, "conv_dgrad")

    def describe(self):
        return "%dx%d/s%d p%d %d->%d" % (self.R, self.S, self.stride, self.pad, self.Ci, self.Co)

    def wgrad(self, dy, x, dw, pro=None):
        """accumulates into dw (fp32 OIHW [Co,Ci,R,S]), issued now (see wgrad_spec)"""
        run_specs([self.wgrad_spec(dy, x, dw, pro=pro)])
        return dw

    def wgrad_spec(self, dy, x, dw, pro=None):
        """accumulates into dw (fp32 OIHW [Co,Ci,R,S]); dy must be dense [N,Ho,Wo,Co_p].  pro as in forward().
        (Two specs sharing a launch use the FIRST one's workspace for both slab regions.)"""
        N, Ho, Wo, Cd = dy.shape
        assert dy.is_contiguous() and Cd == self.Co_p and x.shape[3] == self.Ci_p
        assert dw.dtype == torch.float32 and dw.is_contiguous()
        a = FsWgradArgs()
        a.dy, a.x, a.dw, a.ktab = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), self.ktab_w.data_ptr()
        a.sN, a.sH, a.sW = _nhwc_strides(x)
        a.Hs, a.Ws, a.Hd, a.Wd = x.shape[1], x.shape[2], Ho, Wo
        a.M, a.Cd = N * Ho * Wo, Cd
        a.Co, a.Ci, a.R, a.S = self.Co, self.Ci, self.R, self.S
        a.stride, a.pad, a.ncolgroups, a.pix_per_split = self.stride, self.pad, self.ncolgroups, 0
        ws = wgrad_workspace(dy.device)
        a.workspace, a.workspace_elems = ws.data_ptr(), ws.numel()
        a.x_bytes, a.use_halo = _span_bytes(x), 1
        if pro is not None:
            pst, prelu = pro
            a.pro_a, a.pro_b, a.pro_relu = pst.scale.data_ptr(), pst.shift.data_ptr(), int(prelu)
            a.pro_group_imgs = N // pst.groups if pst.groups > 1 else 0
        flops = 2.0 * a.M * self.Co * self.R * self.S * self.Ci
        return Spec(["conv_wgrad"], a, self.code, "conv_wgrad", flops,
                    lambda: "wgrd %s dy[%d,%d,%d,%d]" % (self.describe(), N, Ho, Wo, Cd), dw, "conv_wgrad")
