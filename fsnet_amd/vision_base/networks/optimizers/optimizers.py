"""build_optimizer with the reference's signature (vision_base/networks/optimizers/optimizers.py:4-11).
'adam' returns FusedAdam: torch.optim.Adam semantics (bias correction, eps outside the sqrt, L2 weight
decay) executed by the HIP clip+Adam kernel — one launch over the meta-arch's flat parameter arena."""
import torch
import torch.nn as nn
import torch.optim as optim

from fsnet_amd.engine.runtime import RT
from fsnet_amd.hip import ops


class FusedAdam(optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, model=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._model = model
        self._arena_state = None
        self._sumsq = None
        self._step_count_fused = 0

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
        if self._arena_state is None or self._arena_state[0] is not arena:
            m = torch.zeros_like(arena.data)
            v = torch.zeros_like(arena.data)
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
        return self._arena_state[1], self._arena_state[2]

    @torch.no_grad()
    def step(self, closure=None, max_norm=None, grad_scale=1.0):
        """max_norm: fused clip_grad_norm_ (joint L2 norm over all parameters) when given."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        arena = self._arena()
        g0 = self.param_groups[0]
        if arena is not None:
            m, v = self._flat_state(arena)
            dev = arena.data.device
            sq = None
            if max_norm is not None and max_norm > 0:
                if self._sumsq is None or self._sumsq.device != dev:
                    self._sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
                self._sumsq.zero_()
                ops.sumsq(arena.grad, self._sumsq)
                sq = self._sumsq
            self._step_count_fused += 1
            ops.adam_step(arena.data, arena.grad, m, v, g0["lr"], g0["betas"][0], g0["betas"][1], g0["eps"],
                          g0["weight_decay"], self._step_count_fused, max_norm=max_norm or 0.0, sumsq_buf=sq,
                          grad_scale=grad_scale)
            for p in arena.params:
                self.state[p]["step"] += 1
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
