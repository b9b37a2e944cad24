"""Process-wide runtime state of the HIP engine: compute dtype policy, the weight-repack epoch,
the flat parameter/gradient arena and the data-parallel context (one process per GPU)."""
import os

import torch


class _Runtime:
    def __init__(self):
        name = os.environ.get("FSNET_AMD_DTYPE", "bf16").lower()
        self.compute_dtype = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32,
                              "float32": torch.float32}[name]
        self.weights_epoch = 0     # bumped whenever parameters change outside torch's version counter
        self.dp = None             # DataParallelContext or None
        self.noise_step = 0        # seed stream for the photometric tie-break noise
        self.tie_noise = os.environ.get("FSNET_AMD_TIE_NOISE", "1") != "0"
        # run the pose chain on a second HIP stream next to the depth chain (they are independent until the
        # loss): measured, ~40 % of the GPU idles in kernel boundaries / tails of the many small launches
        self.overlap = True          # (False: everything on one stream — bench.py's single-stream timing pass, tests)
        # Weight gradients feed nothing downstream in the backward pass: every chain hands them to a companion stream in
        # batches of `wgrad_flush` layers (cross-stream edges are not free — host time eagerly, barrier packets in a
        # hipGraph — hence batches rather than one fork per layer).  wgrad_streams = 0 (inline) is what data parallelism
        # uses by default and what tests may set.
        self.wgrad_streams = 2
        self.wgrad_flush = 8
        # the step runs the same encoder architecture on each of two streams (set per step by MonoDepthMeta.forward_train):
        # the chains are then about equally long, and the main chain's companion — which has the depth decoder's weight
        # gradients on top of the encoder's — ends the backward ~0.3 ms after everything else.  Batches of 4 instead of 8,
        # and from layer2 on the main chain's encoder leaves half of each batch to the pose chain's stream
        # (nets.flush_deferred).  Same box, replayed steps: ResNet-18 + ResNet-18 bf16 5.26 -> 5.20 ms, ResNet-50 +
        # ResNet-50 unchanged (18.2); sharing the whole encoder's batches costs that configuration 2 %, sharing with one
        # chain only (fisheye) 8 %: the shares then run beside the depth chain instead of after it.
        self.even_chains = False
        self.wgrad_flush_even = 4
        self.wgrad_balance = 1
        # inside a capture: issue what is handed over only once the chain has captured its next kernel, so that this kernel
        # is its last node's FIRST successor in the graph, and separate the two chains' first kernels by an empty launch
        # (nets.flush_deferred, nets.chain_begins: the executor's stream assignment follows the edge order)
        self.wgrad_late = True
        self._nop = {}
        # hand over what is pending when an encoder's backward reaches its stem (see EncoderPass.backward): with the two-lane
        # pass only (follows `lanes`)
        self.stem_flush = False
        # the pose encoder's image pairs as one stacked pass with per-pair BatchNorm statistics
        self.batch_pose_pairs = True     # (False: one call per pair, the reference's pattern — tests/test_pose_pairs_gpu.py)
        # the depth encoder and the stacked pose encoder as the two lanes of ONE pass: every post-stem launch carries
        # both networks' problems (EncoderPass in nets.py; fs_*2 entry points).  0: two passes on two streams (round 1-4).
        # Default ("auto"): two chains at world size 1 — measured there on the same box, 150 replayed steps each
        # (profiles/r05_lanes_ab.txt): two chains 5.59-5.79 ms, two lanes 5.82-6.00 (295 launches instead of 415, but the
        # depth decoder no longer runs beside the pose encoder).  Under data parallelism nobody knows in advance: two lanes
        # issue half the SyncBN exchanges and one chain of collectives instead of two that serialise on the communicator,
        # two chains overlap more — the training hook times both on the ranks it has and keeps the faster
        # (BaseTrainingHook, "encoder-pass autotune"); where a step cannot be captured it stays with two lanes.
        self._lanes_env = os.environ.get("FSNET_AMD_LANES", "auto").lower()
        self._lanes = self._lanes_env not in ("0", "auto")
        self._lanes_override = None     # set by the autotune while / after it runs (auto mode only)
        # (measured, same box: two lanes 6.38 -> 6.29 ms with the hand-over; two chains 6.03 -> 6.58 — there the other
        # chain's launches fill the stem's passes already and the extra cross-stream edge delays the chain)
        self.stem_flush = self._lanes
        # the per-step weight re-pack beside the step's weight-free head (nets.pack_everything_async)
        self.pack_overlap = True
        self._side = {}
        self._held = {}           # raw stream handle -> Stream: every stream the engine created (see new_stream)
        # FSNET_AMD_MARKS=1: device-clock marks along the step (mark() below), read back with marks_report()
        self._marks = {} if os.environ.get("FSNET_AMD_MARKS", "0") != "0" else None
        self._mark_buf = None

    @property
    def lanes(self):
        return self._lanes

    @lanes.setter
    def lanes(self, v):
        """True / False: explicit; "auto": by world size (resolve_lanes)"""
        self._lanes_override = None
        if isinstance(v, str) and v.lower() == "auto":
            self._lanes_env, self._lanes = "auto", False
        else:
            self._lanes_env, self._lanes = ("1" if v else "0"), bool(v)
        self.stem_flush = self._lanes

    @property
    def lanes_auto(self):
        return self._lanes_env == "auto"

    @property
    def encoder_pass_ms(self):
        """{"chains": ms, "lanes": ms, "chosen": ...} once the autotune has run for the current data-parallel context"""
        return getattr(self.dp, "encoder_pass_ms", None)

    @encoder_pass_ms.setter
    def encoder_pass_ms(self, v):
        self.dp.encoder_pass_ms = v

    def override_lanes(self, v):
        """auto mode: the autotune's current / final choice (None: back to the world-size rule)"""
        self._lanes_override = None if v is None else bool(v)
        self.resolve_lanes()

    def resolve_lanes(self):
        """FSNET_AMD_LANES=auto: decided when the first training forward knows the world size"""
        if self._lanes_env == "auto":
            if self._lanes_override is not None and self.dp is not None:      # (the autotune's: data parallel only)
                self._lanes = self._lanes_override
            else:
                import torch.distributed as dist
                multi = self.dp is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
                self._lanes = bool(multi)
            self.stem_flush = self._lanes
        return self._lanes

    def mark(self, name):
        """debugging: stamp the device clock on the current stream (a graph node under capture)"""
        if self._marks is None or not torch.cuda.is_available():
            return
        from ..hip.binding import check, lib, stream_ptr
        if self._mark_buf is None:
            self._mark_buf = torch.zeros(512, dtype=torch.int64, device="cuda")
        idx = self._marks.get(name)
        if idx is None:
            idx = self._marks[name] = len(self._marks)
        check(lib.fs_debug_timestamp(self._mark_buf.data_ptr() + 8 * idx, stream_ptr()), "fs_debug_timestamp")

    def marks_report(self):
        """[(name, ms since the earliest mark)] of the last executed step, sorted by time"""
        if not self._marks:
            return []
        torch.cuda.synchronize()
        t = self._mark_buf.cpu().tolist()
        rows = [(n, t[i]) for n, i in self._marks.items() if t[i] > 0]
        t0 = min(v for _, v in rows)
        return sorted(((n, (v - t0) / 1e5) for n, v in rows), key=lambda r: r[1])

    def nop_buffer(self, device):
        """the word nop() writes to.  The training hook asks for it BEFORE it opens a capture: nop() only ever runs inside
        captures, and an allocation there would belong to that graph's pool (and a zero-filled one would add a fill node)."""
        from ..hip.ops import _indexed
        device = _indexed(device)
        t = self._nop.get(device)
        if t is None:
            t = self._nop[device] = torch.empty(1, dtype=torch.int64, device=device)
        return t

    def nop(self, device):
        """one empty-handed launch on the current stream (a node for the graph executor's stream assignment)"""
        from ..hip.binding import check, lib, stream_ptr
        check(lib.fs_debug_timestamp(self.nop_buffer(device).data_ptr(), stream_ptr()), "fs_debug_timestamp")

    def new_stream(self, device):
        """a HIP stream no other part of the engine holds.  torch.cuda.Stream() hands out a pool of 32 streams per
        device round-robin: the 33rd request aliases the first, and two roles of one step (capture stream, pose
        chain, a companion) landing on one stream turns a fork/join into a self-wait inside the capture (seen as a
        segfault in hipGraphLaunch once a process had built many hooks)."""
        for _ in range(64):
            s = torch.cuda.Stream(device=device)
            if s.cuda_stream not in self._held:
                self._held[s.cuda_stream] = s
                return s
        raise RuntimeError("no free HIP stream in torch's pool (every pooled stream is held by the engine)")

    def release_stream(self, handle):
        """give a stream from new_stream() back (its owner is gone)"""
        self._held.pop(handle, None)

    def pack_stream(self, device):
        """stream of the per-step weight re-pack (nets.pack_everything_async)"""
        key = (device, "pack")
        s = self._side.get(key)
        if s is None:
            s = self._side[key] = self.new_stream(device)
        return s

    def side_stream(self, device):
        s = self._side.get(device)
        if s is None:
            s = self._side[device] = self.new_stream(device)
        return s

    def companion_stream(self, device, cur=None):
        """stream that runs weight-gradient kernels next to the chain stream `cur` (default: the current one)"""
        if cur is None:
            cur = torch.cuda.current_stream(device)
        key = (device, cur.cuda_stream, "wgrad")
        s = self._side.get(key)
        if s is None:
            s = self._side[key] = self.new_stream(device)
        return cur, s

    def is_side(self, stream):
        s = self._side.get(stream.device)
        return s is not None and s.cuda_stream == stream.cuda_stream

    def set_compute_dtype(self, dtype):
        if isinstance(dtype, str):
            dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[dtype.lower()]
        assert dtype in (torch.bfloat16, torch.float32)
        self.compute_dtype = dtype

    def bump_weights(self):
        self.weights_epoch += 1


RT = _Runtime()


def require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            "%s: fsnet_amd runs on MI355X through libfsnet_hip.so only — there is no CPU path. "
            "Move the module and its inputs to a ROCm device (.cuda())." % what)


class ParamArena:
    """Flat fp32 storage for all parameters of a meta-arch (+ a same-shaped gradient arena), so that
    clip + Adam is one launch and the data-parallel all-reduce works on a few large contiguous
    buckets.  Parameters keep their identity (state_dict names, optimizer references): only
    `.data` / `.grad` are re-pointed to views."""

    def __init__(self, named_params, device):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        sizes = [p.numel() for p in self.params]
        # 16-byte aligned slots so vector loads in the optimizer never straddle tensors
        self.offsets = []
        off = 0
        for s in sizes:
            self.offsets.append(off)
            off += (s + 3) // 4 * 4
        self.total = off
        self.data = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.views = []
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                v = self.data[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                g = self.grad[o:o + p.numel()].view(p.shape)
                self.views.append(g)
                p.grad = g
        self._by_id = {id(p): i for i, p in enumerate(self.params)}
        self._zero_pending = False
        if self.grad.is_cuda:
            from ..hip import ops
            ops.register_prezero(self, ParamArena._prezero, self.grad.device)
        RT.bump_weights()

    def owns(self, p):
        return id(p) in self._by_id

    def intact(self, full=False):
        """False if something (e.g. module.to(), load with assign) replaced the parameter storage.  Per step only the
        first and last parameter are looked at (whole-model moves: .to(), .float()); full=True checks every one."""
        base = self.data.data_ptr()
        if not full:
            return (self.params[0].data.data_ptr() == base + 4 * self.offsets[0]
                    and self.params[-1].data.data_ptr() == base + 4 * self.offsets[-1])
        return all(p.data.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))

    def attach_grads(self, zero=True):
        if zero:
            self.grad.zero_()
            self._zero_pending = False
        for p, g in zip(self.params, self.views):
            p.grad = g

    def zero_grads(self, lazy=False):
        """lazy (the training hook): the memset joins the step's one scratch-zeroing launch on the re-pack stream
        (nets.pack_everything_async -> ops.prezero_all) instead of a 107 MB fill at the head of the main stream; whoever
        starts the step calls flush_zero() where that launch did not happen"""
        if lazy and self.grad.is_cuda:
            self._zero_pending = True
            self.attach_grads(zero=False)
        else:
            self.attach_grads(zero=True)

    def _prezero(self):
        if not self._zero_pending:
            return []
        self._zero_pending = False
        return [self.grad]

    def flush_zero(self):
        if self._zero_pending:
            self.grad.zero_()
            self._zero_pending = False

    def slice_of(self, params):
        """[lo, hi) element range of the arena covering the given parameters (must be contiguous)."""
        idx = sorted(self._by_id[id(p)] for p in params)
        assert idx == list(range(idx[0], idx[-1] + 1)), "parameters are not contiguous in the arena"
        lo = self.offsets[idx[0]]
        last = idx[-1]
        hi = self.offsets[last] + (self.params[last].numel() + 3) // 4 * 4
        return lo, hi


_ARENAS = []


def register_arena(a):
    _ARENAS.append(a)


def grad_of(p):
    """Gradient buffer to accumulate into (kernels write/accumulate through raw pointers); None for a frozen
    parameter (requires_grad False: frozen stages, resnet.py:179-193) — its kernels are skipped."""
    if not p.requires_grad:
        return None
    if p.grad is None:
        for a in _ARENAS:
            if a.owns(p):
                a.attach_grads(zero=True)   # optimizer.zero_grad(set_to_none=True) happened: one memset
                return p.grad
        p.grad = torch.zeros_like(p)
    return p.grad
