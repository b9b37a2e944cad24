"""One optimisation step, with the reference hook's constructor and call signature
(vision_base/pipeline_hooks/train_val_hooks/base_training_hooks.py:9-49): zero_grad -> H2D -> forward ->
loss.backward() -> clip_grad_norm_ -> optimizer.step().  With the HIP FusedAdam the zero/clip/step
collapse to one memset + two kernels over the flat arena, with no host synchronisation in the step.

hipGraph replay: the step launches several hundred small kernels on two streams and the host needs 8-10 ms to issue them,
about what the GPU needs to run them.  After `graph_warmup` eager steps the hook captures the whole step
(zero-grad memset, weight re-pack, both forward/backward chains, clip + Adam) into one hipGraph on a private
stream and replays it; the per-step scalars (Adam step count, learning rate, tie-break noise seed) live in
device memory, so a replay is a real step.  Inputs are copied into static device buffers; the returned
tensors are the graph's static outputs, valid until the next call (loss_dict entries are cloned when a logger
is attached).  Eager execution stays in use for non-fused optimizers, whenever the batch signature changes, and for
data-parallel steps whose collectives cannot be captured (engine/dataparallel.py).

Encoder-pass autotune (data parallel, FSNET_AMD_LANES=auto): the depth and the pose encoder run either as two chains on
two streams or as the two lanes of one pass (engine/nets.py, EncoderPass).  Which is faster depends on what a SyncBN
exchange costs on the ranks at hand — the lanes issue half as many, the chains overlap more — so the hook captures the
step both ways, times `tune_steps` replays of each with device events (real training steps, every rank in lockstep), the
ranks exchange their timings through the process group's store, and everybody keeps the arrangement whose slowest rank
was fastest (RT.encoder_pass_ms holds every figure).  Where the weight gradients run under data parallelism — inline on
their chain, on companion streams, the depth decoder's at the tail of the pose chain (engine/dataparallel.py) — is timed
the same way: the candidates are the combinations of what was left on "auto"."""
import os
import weakref

import torch

from fsnet_amd.engine.runtime import RT
from fsnet_amd.vision_base.utils.timer import profile


# ROCm 7.0 / 7.2: destroying a hipGraphExec can leave graphs instantiated LATER in the process with too few internal streams —
# hip::Graph::UpdateStreams reads past the exec's parallel-stream vector on their first hipGraphLaunch and the process
# segfaults (native backtrace: tools/probes/segv_bt.sh; reproduced with three test files in sequence, gone as soon as no exec
# is destroyed).  Captured steps are therefore PARKED, not destroyed, when a hook lets go of them — a changed batch signature,
# the autotune's losing arrangements, the hook itself going away: this list owns every captured graph until the process ends
# (one graph's private pool is the step's activations: 2.3 GB at the benchmark size; a training run retires a handful).
# Graphs captured under data parallelism hold state of the context's RCCL communicator (a graph outliving its communicator
# aborts the process when it is finally freed): those are owned by the data-parallel context and destroyed with it, before
# the communicator goes (engine/dataparallel.py: park_graph / close) — one context per process lifetime in a training run;
# the RCCL tests run each context in a process of its own (tests/test_dp_gpu.py).
_PARKED = []


def _own_graph(graph):
    if RT.dp is not None:
        RT.dp.park_graph(graph)
    else:
        _PARKED.append(graph)


class BaseTrainingHook(object):
    def __init__(self, tensor_keys=None, clip_gradients=None, use_graph=None, graph_warmup=3, **kwargs):
        self.tensor_keys = tensor_keys
        self.clip_gradients = clip_gradients
        if use_graph is None:
            use_graph = os.environ.get("FSNET_AMD_GRAPH", "1") != "0"
        self.use_graph = bool(use_graph)
        # data-parallel steps are captured too when every collective of the step runs on the engine's direct RCCL
        # communicator and its start-up capture self-test passed on all ranks (engine/dataparallel.py, rccl_direct.py):
        # SyncBN exchanges and gradient buckets are then graph nodes and no torch.distributed work object exists while
        # the capture is open.  FSNET_AMD_GRAPH_DP=0 keeps them eager.
        self.graph_dp = os.environ.get("FSNET_AMD_GRAPH_DP", "1") != "0"
        self.graph_warmup = int(graph_warmup)   # eager steps before the capture (at least 2: see __call__)
        self.graph_captures = 0
        self._g = None            # dict(graph, sig, static, output, stream, ...) once captured
        self._g_sig = None
        self._g_eager = 0         # eager steps seen with the current signature
        self._g_stream = None
        self._seed = None
        self.graph_replays = 0
        # encoder-pass autotune: None = not looked at yet, dict = running, False = finished or not applicable
        self._tune = None
        self.tune_steps = int(os.environ.get("FSNET_AMD_TUNE_STEPS", "10"))

    # ------------------------------------------------------------------ encoder-pass autotune (data parallel)
    @property
    def tune_done(self):
        return self._tune is False

    @staticmethod
    def _tune_name(cand):
        return ("lanes" if cand[0] else "chains") + {"inline": "", "tail": "+tail", "companion": "+companions"}[cand[1]]

    def _tune_init(self, inner, data):
        """first call: is there a choice to make?  Only under data parallelism, for what was left on "auto" — the encoder
        arrangement (FSNET_AMD_LANES) and where the weight gradients run (FSNET_AMD_DP_WGRAD, engine/dataparallel.py) —
        once per data-parallel context"""
        import torch.distributed as dist
        multi = RT.dp is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        if not (multi and RT.encoder_pass_ms is None and self.tune_steps > 0):
            self._tune = False
            return
        possible = getattr(inner, "lanes_possible", None)
        lanes = [False, True] if (RT.lanes_auto and possible is not None and possible(None)) else [None]
        wenv = os.environ.get("FSNET_AMD_DP_WGRAD", "auto").lower()
        wgrad = ["inline", "tail", "companion"] if wenv == "auto" else [wenv]
        cands = [(l, w) for l in lanes for w in wgrad]
        if len(cands) < 2:
            self._tune = False
            return
        self._tune = dict(phase=0, cands=cands, n=0, ms={}, graphs={}, ev0=None, t0=None)
        self._tune_apply(cands[0])

    def _tune_apply(self, cand):
        lanes, wgrad = cand
        if lanes is not None:
            RT.override_lanes(lanes)
        if RT.dp is not None:                # (created by the first training forward)
            RT.dp.wgrad_mode = wgrad

    def _tune_abort(self):
        """a capture failed under the autotune's feet (the communicator is being re-made): back to the defaults"""
        self._tune["graphs"].clear()
        self._tune = False
        RT.override_lanes(None)
        if RT.dp is not None:
            RT.dp.wgrad_mode = "inline" if RT.dp.wgrad_env == "auto" else RT.dp.wgrad_env

    def _tune_phase_end(self, t, ms):
        """`ms`: this rank's time per step in the arrangement that just ran"""
        cands = t["cands"]
        t["ms"][t["phase"]] = ms
        t["graphs"][t["phase"]] = (self._g, self._g_sig)
        if t["phase"] + 1 < len(cands):
            # the next calls warm up (and capture) the next arrangement; the graphs so far stay alive beside it
            t["phase"], t["n"] = t["phase"] + 1, 0
            self._g, self._g_sig, self._g_eager = None, None, 0
            self._tune_apply(cands[t["phase"]])
            return
        rows = RT.dp.gather_floats([t["ms"][i] for i in range(len(cands))])    # every rank's figures, the same list everywhere
        worst = [max(r[i] for r in rows) for i in range(len(cands))]
        best = min(range(len(cands)), key=lambda i: (worst[i], i))
        RT.encoder_pass_ms = dict({self._tune_name(c): round(w, 4) for c, w in zip(cands, worst)},
                                  chosen=self._tune_name(cands[best]), steps=self.tune_steps, ranks=len(rows),
                                  timed="hipgraph replays" if self._g is not None else "eager steps")
        self._tune_apply(cands[best])
        self._g, self._g_sig = t["graphs"][best]
        t["graphs"].clear()                                     # (the other graphs go: the device is idle)
        self._tune = False

    def _tune_eager_step(self, data, meta_arch, optimizer, arena, fused, meta, logger):
        """data-parallel steps that are not captured (gloo, a failed communicator self-test): two warm-up steps, then
        `tune_steps` eager steps per arrangement between two device synchronisations"""
        import time
        t = self._tune
        if t["n"] == 2:
            torch.cuda.synchronize()
            t["t0"] = time.perf_counter()
        output = self._eager_step(data, meta_arch, optimizer, arena, fused, meta, logger)
        t["n"] += 1
        if t["n"] >= 2 + self.tune_steps:
            torch.cuda.synchronize()
            self._tune_phase_end(t, (time.perf_counter() - t["t0"]) * 1e3 / self.tune_steps)
        return output

    # ------------------------------------------------------------------ graph path
    def _signature(self, data, meta_arch, optimizer):
        sig = [id(meta_arch), id(optimizer), self.clip_gradients]
        # Adam's betas / eps / weight decay are baked into the captured launch (only lr and the step count live in
        # device memory): changing them re-captures
        for g in getattr(optimizer, "param_groups", []):
            sig.append((tuple(g.get("betas", ())), g.get("eps"), g.get("weight_decay")))
        for k, v in data.items():
            if isinstance(v, torch.Tensor):
                sig.append((k, tuple(v.shape), v.dtype))
        return tuple(sig)

    def _graph_ok(self, meta_arch, optimizer, arena, fused):
        if not (self.use_graph and fused and arena is not None and torch.cuda.is_available()):
            return False
        if RT.dp is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()
                                 and torch.distributed.get_world_size() > 1):
            # (RT.dp is created by the first forward: the warm-up steps run before this can say yes)
            if not (self.graph_dp and RT.dp is not None and RT.dp.capturable):
                return False
        if not next(meta_arch.parameters()).is_cuda:
            return False
        return True

    def _stage(self, data, static):
        """incoming batch -> static device buffers (H2D or D2D on the current stream)"""
        from fsnet_amd.hip import ops
        ops.copy_multi([(static[k], v) for k, v in data.items() if isinstance(v, torch.Tensor) and k in static])

    def _backward(self, loss):
        """loss.backward() with the seed gradient handed in (autograd otherwise launches a fill for its ones_like — a node of
        its own between the loss and its backward, the one stretch of the step nothing runs beside)"""
        loss = loss if loss.dim() == 0 else loss.mean()
        seed = self._seed
        if seed is None or seed.device != loss.device or seed.dtype != loss.dtype:
            seed = self._seed = torch.ones((), dtype=loss.dtype, device=loss.device)
        loss.backward(gradient=seed)

    def _eager_step(self, data, meta_arch, optimizer, arena, fused, meta, logger):
        if arena is not None:
            arena.zero_grads(lazy=True)       # (zeroed with the step's scratch at the head of the forward: _begin_train)
        else:
            optimizer.zero_grad()
        stage = getattr(getattr(meta_arch, "module", meta_arch), "stage_step_inputs", None)
        if stage is not None:
            stage(data)        # non-tensor inputs (fisheye calibrations), before the H2D move
        for key in data:
            if isinstance(data[key], torch.Tensor):
                if self.tensor_keys is None or key in self.tensor_keys:
                    data[key] = data[key].cuda(non_blocking=True).contiguous()
        output = meta_arch(data, meta)
        if logger is not None:
            logger.update(output['loss_dict'])
            logger.update_hm(output.get('hm', dict()))
        loss = output['loss']
        self._backward(loss)
        self._optim(meta_arch, optimizer, fused)
        return output

    def _capture(self, data, meta_arch, optimizer, arena, meta, sig):
        dev = next(meta_arch.parameters()).device
        static = {}
        for k, v in data.items():
            if isinstance(v, torch.Tensor):
                static[k] = torch.empty(v.shape, dtype=v.dtype, device=dev).contiguous()
        sdata = dict(data)
        sdata.update(static)
        self._stage(data, static)
        stage = getattr(getattr(meta_arch, "module", meta_arch), "stage_step_inputs", None)
        if stage is not None:
            stage(data)
        optimizer.sync_lr()
        RT.nop_buffer(dev)
        torch.cuda.synchronize(dev)
        if RT.dp is not None:
            from fsnet_amd.engine.rccl_direct import quiesce_watchdog
            quiesce_watchdog(dev)
        dot = os.environ.get("FSNET_AMD_GRAPH_DOT")      # debugging: the captured topology as a DOT file
        graph = torch.cuda.CUDAGraph(keep_graph=True) if dot else torch.cuda.CUDAGraph()
        if dot:
            graph.enable_debug_mode()
        steps_before = optimizer._step_count_fused
        # (with a process group alive its watchdog thread polls events: only this thread's calls may fail the capture)
        mode = "thread_local" if RT.dp is not None else "global"
        with torch.cuda.graph(graph, stream=self._g_stream, capture_error_mode=mode):
            RT.mark("step.start")
            arena.zero_grads(lazy=True)
            output = meta_arch(sdata, meta)
            loss = output['loss']
            RT.mark("loss.fwd.end")
            self._backward(loss)
            RT.mark("bwd.joined")
            grad_scale = RT.dp.finish() if RT.dp is not None else 1.0
            optimizer.step(max_norm=self.clip_gradients, grad_scale=grad_scale)
            RT.mark("step.end")
        if dot:
            graph.instantiate()
            graph.debug_dump(dot)
        # capture records, it does not run: host bookkeeping happened once above, the first replay is that step
        assert optimizer._step_count_fused == steps_before + 1
        self._g = dict(graph=graph, sig=sig, static=static, output=output, arena=arena,
                       stage_meta=stage)
        _own_graph(graph)
        if RT.dp is not None:
            RT.dp.note_graph_owner(self)     # the graph holds RCCL nodes: it must go before the communicator does
            return output                    # data parallel: the ranks agree first (__call__), then _first_replay()
        return self._first_replay()

    def _first_replay(self):
        graph, output = self._g["graph"], self._g["output"]
        graph.replay()
        RT.bump_weights()
        return output

    def _replay(self, data, optimizer):
        g = self._g
        self._stage(data, g["static"])
        if g["stage_meta"] is not None:
            g["stage_meta"](data)
        optimizer.prepare_replay()
        g["graph"].replay()
        optimizer.note_step()
        RT.bump_weights()          # packed MFMA operands are one Adam step behind the arena again
        self.graph_replays += 1
        return g["output"]

    def reset_graph(self):
        self._g, self._g_sig, self._g_eager = None, None, 0
        if self._tune:
            self._tune_abort()

    def _optim(self, meta_arch, optimizer, fused):
        grad_scale = RT.dp.finish() if RT.dp is not None else 1.0
        if fused:
            optimizer.step(max_norm=self.clip_gradients, grad_scale=grad_scale)
        else:
            if grad_scale != 1.0:
                for p in meta_arch.parameters():
                    if p.grad is not None:
                        p.grad.mul_(grad_scale)
            if self.clip_gradients is not None:
                torch.nn.utils.clip_grad_norm_(meta_arch.parameters(), self.clip_gradients)
            optimizer.step()
            RT.bump_weights()

    @profile('Training hook', 0, 100)
    def __call__(self, data, meta_arch, optimizer, writer=None, training_loss_logger=None, global_step=0,
                 epoch_num=0):
        from fsnet_amd.engine.torch_compat import adopt_optimizer
        from fsnet_amd.vision_base.networks.optimizers.optimizers import FusedAdam
        inner = getattr(meta_arch, "module", meta_arch)
        optimizer = adopt_optimizer(optimizer, meta_arch)     # torch.optim.Adam from the reference's build_optimizer
        arena = inner.ensure_arena() if hasattr(inner, "ensure_arena") else None
        fused = isinstance(optimizer, FusedAdam)
        meta = dict(epoch_num=epoch_num, global_step=global_step, is_training=True)
        logger = training_loss_logger

        if self._tune is None:
            self._tune_init(inner, data)
        elif self._tune:
            self._tune_apply(self._tune["cands"][self._tune["phase"]])
        if not self._graph_ok(meta_arch, optimizer, arena, fused):
            if self._tune and RT.dp is not None:        # (RT.dp is created by the first forward: only then is it known)
                return self._tune_eager_step(data, meta_arch, optimizer, arena, fused, meta, logger)
            return self._eager_step(data, meta_arch, optimizer, arena, fused, meta, logger)

        sig = self._signature(data, meta_arch, optimizer)
        if sig != self._g_sig or (self._g is not None and not self._g["arena"] is arena):
            # (the old signature's graph stays parked, see _PARKED: it is not replayed again — objects the model caches per
            # batch geometry, the loss chain's buffers among them, are rebuilt for the new one and the old graph's pointers
            # into them die)
            self._g, self._g_sig, self._g_eager = None, sig, 0
            if self._tune:
                self._tune["n"] = 0          # a timing window does not span a re-capture
        if self._g_stream is None:
            self._g_stream = RT.new_stream(next(meta_arch.parameters()).device)
            weakref.finalize(self, RT.release_stream, self._g_stream.cuda_stream)
        cur = torch.cuda.current_stream()
        if self._g is None and self._g_eager < max(2, self.graph_warmup):
            # warm-up on the capture stream: lazy buffers, per-stream pools and tables exist before the capture
            # (two steps at least: the one-launch weight re-pack table is first built by the second step)
            self._g_eager += 1
            self._g_stream.wait_stream(cur)
            with torch.cuda.stream(self._g_stream):
                output = self._eager_step(data, meta_arch, optimizer, arena, fused, meta, logger)
            cur.wait_stream(self._g_stream)
            return output
        self._g_stream.wait_stream(cur)
        with torch.cuda.stream(self._g_stream):
            if self._g is None:
                steps_before = optimizer._step_count_fused
                failure = None
                try:
                    output = self._capture(data, meta_arch, optimizer, arena, meta, sig)
                except Exception as e:      # something in the step is not capturable here: stay eager, loudly
                    failure = "%s: %s" % (type(e).__name__, e)
                if RT.dp is not None:
                    # Every rank replays or none does: a rank that fell back to eager launches would issue its
                    # collectives from the host while the others replay theirs from a graph, and a rank whose capture
                    # died would not issue them at all.  (MIN all-reduce over torch.distributed, outside any capture.)
                    if not RT.dp.all_agree(failure is None):
                        if failure is None:
                            failure = "the capture failed on another rank"
                        # what this rank (or the failing one) captured must not linger on the communicator the eager
                        # step is about to use
                        torch.cuda.synchronize()
                        RT.dp.reset_direct()
                    if failure is None:
                        output = self._first_replay()
                if failure is None:
                    self.graph_captures += 1
                else:
                    import warnings
                    warnings.warn("fsnet_amd: hipGraph capture of the training step failed (%s); "
                                  "continuing with eager launches" % failure)
                    self.use_graph = False
                    self._g = None
                    if self._tune:
                        self._tune_abort()
                    optimizer._step_count_fused = steps_before
                    torch.cuda.synchronize()
                    output = self._eager_step(data, meta_arch, optimizer, arena, fused, meta, logger)
                    logger = None
            else:
                t = self._tune
                if t and t["n"] == 0:
                    t["ev0"] = torch.cuda.Event(enable_timing=True)
                    t["ev0"].record()
                output = self._replay(data, optimizer)
                if t:
                    t["n"] += 1
                    if t["n"] >= self.tune_steps:
                        ev1 = torch.cuda.Event(enable_timing=True)
                        ev1.record()
                        ev1.synchronize()
                        self._tune_phase_end(t, t["ev0"].elapsed_time(ev1) / self.tune_steps)
            if logger is not None:
                logger.update({k: v.clone() for k, v in output['loss_dict'].items()})
                logger.update_hm(output.get('hm', dict()))
        cur.wait_stream(self._g_stream)
        return output
