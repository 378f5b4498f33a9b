"""Time stepping for right-hand sides the fused kernels cannot hold: an arbitrary ``model=`` module.

The reference's ``ScoreModel`` takes any ``nn.Module`` with the signature ``(t, x, conditional=None)`` as its score
network (flowfusion/diffusion.py:201, 233-238) -- the notebooks plug closed-form scores in -- and leaves the
stepping to torchdiffeq (call sites :631-639, :744-752) or to its own Euler-Maruyama loop (:543-562).  The fused
kernels need the network's weights, so such a model cannot run inside them; what the library still takes over is
the stepping: the same evaluation plans (time grid, stage coefficients: solvers.plan_ode) and the same adaptive
driver (adaptive.Dopri5) as the fused path, with the right-hand side evaluated by the user's module on the GPU and
every stage input / step update / dense-output / error-estimate combination done in ONE pass over the state by
``ff_stage_combine`` (csrc/ff_aux.hip; torchdiffeq spends one elementwise pass per term).  GPU only, like the rest
of the package: CPU tensors raise.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

from . import _native, solvers
from .solvers import FLAG_STEP_END, MAX_SLOTS

# rhs(t [0-dim fp32 tensor on the state's device, real time], y [B, D]) -> (ydot [B, D], div [B] or None)
Rhs = Callable[[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, Optional[torch.Tensor]]]


def _need_gpu(x: torch.Tensor):
    if not x.is_cuda:
        raise RuntimeError(
            "flowfusion_amd integrates on the GPU only: move the model and its inputs to 'cuda' "
            f"(got a tensor on {x.device}); there is no CPU fallback")


def _combine(x: Optional[torch.Tensor], ks, coefs, like: torch.Tensor, x_coef: float = 1.0) -> torch.Tensor:
    """x_coef * x + sum_s coefs[s] * ks[s] into a fresh tensor shaped like `like` (one ff_stage_combine launch)."""
    out = torch.empty_like(like)
    return _native.stage_combine(out, x, [k if k is not None else None for k in ks], coefs, x_coef)


class ModuleStepper:
    """Explicit Runge-Kutta stepping around an external right-hand side (same row semantics as the fused kernel:
    ``y = x + sum cin[s] k[s]``, ``k[slot] = f``, at STEP_END ``x += sum cout[s] k[s]``)."""

    def __init__(self, rhs: Rhs, has_lp: bool):
        self.rhs = rhs
        self.has_lp = has_lp
        self.n_evals = 0

    def _f(self, t_real: float, y: torch.Tensor, sign: float):
        """Solver-time derivative ``sign * f(t_real, y)`` (a decreasing span is solved in reversed time with the
        right-hand side negated, as torchdiffeq does and as the fused kernel's rows encode)."""
        t = torch.tensor(t_real, dtype=torch.float32, device=y.device)
        ydot, div = self.rhs(t, y)
        self.n_evals += 1
        ydot = ydot.detach().to(torch.float32)
        ydot = (-ydot if sign < 0 else ydot).contiguous()
        if self.has_lp:
            div = div.detach().to(torch.float32).reshape(-1)
            div = (-div if sign < 0 else div).contiguous()
        return ydot, (div if self.has_lp else None)

    def run_plan(self, x: torch.Tensor, plan: solvers.EvalPlan):
        """Fixed-grid integration over the rows of an evaluation plan; returns (x_final, dlogp [B] or None)."""
        _need_gpu(x)
        x = x.detach().to(torch.float32).contiguous()
        lp = torch.zeros(x.shape[0], device=x.device) if self.has_lp else None
        ks = [None] * MAX_SLOTS
        kl = [None] * MAX_SLOTS
        cin, cout = plan.cin.tolist(), plan.cout.tolist()
        t_eval, slots, flags = plan.t_eval.tolist(), plan.slot.tolist(), plan.flags.tolist()
        for e in range(len(t_eval)):
            y = _combine(x, ks, cin[e][:MAX_SLOTS], x)
            ks[slots[e]], kl[slots[e]] = self._f(t_eval[e], y, plan.sign)
            if flags[e] & FLAG_STEP_END:
                x = _combine(x, ks, cout[e][:MAX_SLOTS], x)
                if self.has_lp:
                    lp = _combine(lp, kl, cout[e][:MAX_SLOTS], lp)
        return x, lp

    def make_step(self, sign: float):
        """Step function for ``adaptive.Dopri5`` (same contract as ``FusedNet.make_step``): the rows fill the stage
        slots, ``n_aux`` combinations of them come back."""
        def step(y, k1, lp0, kl1, t_rows, cin, slots, tail, use_y, n_aux):
            ks = [None] * MAX_SLOTS
            kl = [None] * MAX_SLOTS
            ks[0], kl[0] = k1, kl1
            ts = (sign * t_rows).tolist()                   # real time of every row
            for i in range(len(ts)):
                yi = _combine(y, ks, cin[i, :MAX_SLOTS].tolist(), y)
                ks[int(slots[i])], kl[int(slots[i])] = self._f(ts[i], yi, sign)
            aux = torch.empty(n_aux, *y.shape, device=y.device, dtype=torch.float32)
            aux_lp = torch.zeros(n_aux, y.shape[0], device=y.device) if self.has_lp else None
            for j in range(n_aux):
                uy = float((use_y >> j) & 1)
                cj = tail[j, :MAX_SLOTS].tolist()
                _native.stage_combine(aux[j], y, ks, cj, uy)
                if self.has_lp:
                    _native.stage_combine(aux_lp[j], lp0 if lp0 is not None else aux_lp[j], kl, cj, uy if lp0 is not None else 0.0)
            return aux, aux_lp
        return step


def solve(rhs: Rhs, x: torch.Tensor, t_span: torch.Tensor, method: str, options, has_lp: bool, atol: float, rtol: float,
          norm_only=()):
    """``odeint(func, state, t_span, method=, atol=, rtol=, options=)`` for an external right-hand side: fixed-grid
    methods (solvers.FIXED_METHODS) and adaptive dopri5.  Returns (y [B, D], dlogp [B] or None) at ``t_span[-1]``."""
    from . import adaptive
    _need_gpu(x)
    stepper = ModuleStepper(rhs, has_lp)
    if method in solvers.ALL_ADAPTIVE:
        t = t_span.detach().to("cpu", torch.float32).double()
        sign = -1.0 if bool(t[0] > t[-1]) else 1.0
        solver = adaptive.make_solver(stepper.make_step(sign), has_lp, rtol, atol, options, norm_only=norm_only, method=method, sign=sign)
        lp0 = torch.zeros(x.shape[0], device=x.device) if has_lp else None
        y, lp = solver.integrate(float(sign * t[0]), float(sign * t[-1]), x.detach().to(torch.float32).contiguous(), lp0)
        return y, lp, {"attempts": solver.n_attempts, "accepted": solver.n_accepted, "evaluations": stepper.n_evals}
    plan = solvers.plan_ode(t_span, method, options, y0=x)
    y, lp = stepper.run_plan(x, plan)
    return y, lp, {"evaluations": stepper.n_evals}


def euler_maruyama(drift: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], g_of_t: Callable[[torch.Tensor], torch.Tensor],
                   x: torch.Tensor, draw, T, epsilon, steps: int, progress=None):
    """The reverse-SDE loop of ``sample_sde`` (flowfusion/diffusion.py:539-563) for an external drift
    ``f - g^2 score``: per step one module call, ``x_mean = x + f dt`` and ``x = x_mean + g sqrt(-dt) dw`` as one
    ff_stage_combine each.  Stops at the first step whose state holds a NaN and returns that step's mean, like the
    reference (:560-563)."""
    _need_gpu(x)
    ts, dt = solvers.plan_euler_maruyama(T, epsilon, steps)
    n = int(ts.numel())
    if n == 0:
        raise RuntimeError("sample_sde: T < epsilon, no step to take")
    g = g_of_t(ts).reshape(-1)                              # host copy of the SDE: g depends on time only
    gn = (g * (-dt) ** (1.0 / 2.0)).tolist()
    dtf = float(dt)
    x = x.detach().to(torch.float32).contiguous()
    x_mean = x
    for i in range(n):
        t = torch.ones(x.shape[0], device=x.device) * ts[i].to(x.device)           # per-sample vector, as the reference (:540)
        f = drift(t, x).detach().to(torch.float32).contiguous()
        dw = draw(x)
        x_mean = _combine(x, [f], [dtf], x)
        x = _combine(x_mean, [dw], [gn[i]], x)
        if bool(torch.isnan(x).any()):
            print("Diffusion is not stable, NaN were produced. Stopped sampling.")
            break
        if progress is not None:
            progress(i + 1, n)
    return x_mean
