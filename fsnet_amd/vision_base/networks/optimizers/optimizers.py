"""build_optimizer with the reference's signature (vision_base/networks/optimizers/optimizers.py:4-11).
'adam' returns FusedAdam: torch.optim.Adam semantics (bias correction, eps outside the sqrt, L2 weight
decay) executed by the HIP clip+Adam kernel — one launch over the meta-arch's flat parameter arena."""
import torch
import torch.nn as nn
import torch.optim as optim

from fsnet_amd.engine.nets import join_companions_final
from fsnet_amd.engine.runtime import RT
from fsnet_amd.hip import ops


class FusedAdam(optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, model=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._model = model
        self._arena_state = None
        self._readopt = False
        self._sumsq = None
        self._clean_sq = False
        self._step_count_fused = 0
        self._step_buf = None      # device-resident step count / lr: the step kernel is hipGraph-replayable
        self._lr_buf = None
        self._lr_host = None

    # -- flat path -------------------------------------------------------------------------------
    def _arena(self):
        m = self._model
        m = getattr(m, "module", m)
        arena = getattr(m, "_arena", None)
        if arena is None or len(self.param_groups) != 1:
            return None
        ps = self.param_groups[0]["params"]
        if len(ps) != len(arena.params) or any(a is not b for a, b in zip(ps, arena.params)):
            return None
        return arena

    def _flat_state(self, arena):
        if self._arena_state is None or self._arena_state[0] is not arena or self._readopt:
            if self._arena_state is not None and self._arena_state[0] is arena:
                m, v = self._arena_state[1].zero_(), self._arena_state[2].zero_()   # same storage: graphs stay valid
            else:
                m = torch.zeros_like(arena.data)
                v = torch.zeros_like(arena.data)
            self._readopt = False
            step0 = 0
            for p, o in zip(arena.params, arena.offsets):   # adopt state loaded from a checkpoint
                st = self.state.get(p, None)
                if st and "exp_avg" in st:
                    m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                    v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                    step0 = int(st["step"])
                # expose per-parameter views so state_dict() keeps torch.optim.Adam's format
                self.state[p] = {"step": torch.tensor(float(step0)), "exp_avg": m[o:o + p.numel()].view(p.shape),
                                 "exp_avg_sq": v[o:o + p.numel()].view(p.shape)}
            self._arena_state = (arena, m, v)
            self._step_count_fused = step0
            if self._step_buf is not None:       # keep the pointer (a captured graph holds it), reset the value
                self._step_buf.fill_(step0)
        return self._arena_state[1], self._arena_state[2]

    def _prezero(self):
        """(ops.prezero_all) the gradient-norm accumulator is about to be zeroed with the step's scratch"""
        if self._sumsq is None:
            return []
        self._clean_sq = True
        return [self._sumsq_buf]

    @torch.no_grad()
    def step(self, closure=None, max_norm=None, grad_scale=1.0):
        """max_norm: fused clip_grad_norm_ (joint L2 norm over all parameters) when given."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        arena = self._arena()
        g0 = self.param_groups[0]
        join_companions_final()        # weight-gradient kernels still in flight on companion streams
        if arena is not None:
            if g0["weight_decay"] and any(not p.requires_grad for p in arena.params):
                # frozen parameters keep a zero gradient in the arena, which Adam turns into a zero update — but L2
                # weight decay would still shrink them (torch.optim.Adam skips parameters without a gradient)
                raise NotImplementedError("FusedAdam: weight_decay with frozen parameters in the arena")
            m, v = self._flat_state(arena)
            dev = arena.data.device
            if self._step_buf is None or self._step_buf.device != dev:
                self._step_buf = torch.full((1,), self._step_count_fused, dtype=torch.int32, device=dev)
                self._lr_buf = torch.full((1,), float(g0["lr"]), dtype=torch.float32, device=dev)
                self._lr_host = float(g0["lr"])
            self.sync_lr()
            sq = None
            if max_norm is not None and max_norm > 0:
                if self._sumsq is None or self._sumsq.device != dev:
                    # (16 bytes: the step's one-launch scratch zeroing takes 16-byte aligned spans)
                    self._sumsq_buf = torch.zeros(2, dtype=torch.float64, device=dev)
                    self._sumsq = self._sumsq_buf[:1]
                    self._clean_sq = False
                    ops.register_prezero(self, lambda o: o._prezero(), dev)
                if self._clean_sq:
                    from fsnet_amd.engine.nets import join_pack
                    join_pack(dev)
                else:
                    self._sumsq.zero_()
                self._clean_sq = False
                ops.sumsq(arena.grad, self._sumsq, step_counter=self._step_buf)   # also bumps the device step count
                sq = self._sumsq
            else:
                ops.counter_incr(self._step_buf)
            ops.adam_step(arena.data, arena.grad, m, v, g0["lr"], g0["betas"][0], g0["betas"][1], g0["eps"],
                          g0["weight_decay"], 1, max_norm=max_norm or 0.0, sumsq_buf=sq,
                          grad_scale=grad_scale, step_buf=self._step_buf, lr_buf=self._lr_buf)
            self.note_step()
        else:
            params = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
            sq = None
            if max_norm is not None and max_norm > 0 and params:
                sq = torch.zeros(1, dtype=torch.float64, device=params[0].device)
                for p in params:
                    ops.sumsq(p.grad, sq)
            for g in self.param_groups:
                for p in g["params"]:
                    if p.grad is None:
                        continue
                    st = self.state[p]
                    if not st:
                        st["step"] = torch.tensor(0.0)
                        st["exp_avg"] = torch.zeros_like(p)
                        st["exp_avg_sq"] = torch.zeros_like(p)
                    st["step"] += 1
                    ops.adam_step(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], g["lr"], g["betas"][0],
                                  g["betas"][1], g["eps"], g["weight_decay"], int(st["step"]),
                                  max_norm=max_norm or 0.0, sumsq_buf=sq, grad_scale=grad_scale)
        RT.bump_weights()   # parameters changed through raw pointers: conv operands must be re-packed
        return loss

    def note_step(self):
        """host-side bookkeeping of one (eager or graph-replayed) fused step"""
        self._step_count_fused += 1

    def _refresh_steps(self):
        if self._arena_state is not None:
            for p in self._arena_state[0].params:
                self.state[p]["step"].fill_(float(self._step_count_fused))

    def state_dict(self):
        self._refresh_steps()      # per-parameter "step" entries are only materialised when someone looks
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._readopt = True       # re-adopt moments / step count from the loaded per-parameter state

    def prepare_replay(self):
        """before a captured step is replayed: adopt freshly loaded state in place, push the current lr"""
        arena = self._arena()
        if self._readopt and arena is not None:
            self._flat_state(arena)
        self.sync_lr()

    def sync_lr(self):
        """push a scheduler-changed learning rate to the device scalar the (possibly captured) kernel reads"""
        lr = float(self.param_groups[0]["lr"])
        if self._lr_buf is not None and lr != self._lr_host:
            self._lr_buf.fill_(lr)
            self._lr_host = lr

    def grad_norm(self):
        """sqrt of the last fused sum of squares (device tensor; no host sync)."""
        return None if self._sumsq is None else self._sumsq.sqrt()


def build_optimizer(model: nn.Module, name, **kwargs):
    if name.lower() == 'adam':
        return FusedAdam(model.parameters(), model=model, **kwargs)
    if name.lower() == 'sgd':
        return optim.SGD(model.parameters(), **kwargs)
    if name.lower() == 'adamw':
        return optim.AdamW(model.parameters(), **kwargs)
    raise NotImplementedError(name)
