"""Adaptive Dormand-Prince 5(4) on top of the fused kernel -- the reference's *default* solver.

Every ODE solve in the reference defaults to ``method="dopri5"`` (diffusion.py:572, 649, 763;
flow.py:313 and the argument-less ``odeint`` call in ``sample``, flow.py:299-303), i.e.
torchdiffeq's adaptive Dormand-Prince with one step size for the whole batch: the error norm is
taken over the entire state (for a tuple state: the maximum of the per-component RMS norms), so
every sample walks the same time grid.  That global decision is kept on the host (a few scalar
reductions per attempted step); everything per sample runs in the fused kernel, one launch per
attempted step: six network evaluations (stages 2..7, the first one is the FSAL derivative handed
back in), after which the new state, the last stage, the dense-output midpoint and the error
estimate leave the chip as linear combinations of the stage slots (ff_ode_args.aux_*).

torchdiffeq (>=0.2.5,<0.3.0) is not available offline; this file restates its published algorithm
(rk_common.py: ``_select_initial_step``, ``_runge_kutta_step``, ``_compute_error_ratio``,
``_optimal_step_size``, 4th-order dense output ``_interp_fit`` / ``_interp_evaluate``, min/max step
handling) including where it computes in the state's fp32 and where in float64 time.  Parity with
the real package is unpinned; the CPU oracle carries an independent restatement that
the GPU tests compare against.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

# Dormand-Prince 5(4) tableau as torchdiffeq writes it (dopri5.py)
ALPHA = (1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
BETA = (
    (1 / 5,),
    (3 / 40, 9 / 40),
    (44 / 45, -56 / 15, 32 / 9),
    (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
    (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
    (35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84),
)
C_SOL = (35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0)
C_ERROR = (
    35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
    -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1.0 / 60.0,
)
C_MID = (
    6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
    187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2,
)
ORDER = 5
SAFETY, IFACTOR, DFACTOR = 0.9, 10.0, 0.2


class EmbeddedTableau:
    """An embedded explicit Runge-Kutta pair with the first-same-as-last property, as torchdiffeq's
    ``_ButcherTableau`` + ``c_mid`` + ``order``: stage i+1 at ``t0 + alpha[i] dt`` on ``y0 + dt sum_j beta[i][j] k_j``;
    the last stage is evaluated at (t1, y1), so it is the next step's first stage."""

    def __init__(self, name, order, alpha, beta, c_sol, c_error, c_mid):
        self.name, self.order = name, order
        self.alpha, self.beta, self.c_sol, self.c_error, self.c_mid = alpha, beta, c_sol, c_error, c_mid
        self.stages = len(c_sol)                      # including the FSAL stage


TABLEAUX = {
    "dopri5": EmbeddedTableau("dopri5", 5, ALPHA, BETA, C_SOL, C_ERROR, C_MID),
    # Bogacki-Shampine 3(2) (torchdiffeq bosh3.py)
    "bosh3": EmbeddedTableau("bosh3", 3, (1 / 2, 3 / 4, 1.0), ((1 / 2,), (0.0, 3 / 4), (2 / 9, 1 / 3, 4 / 9)),
                             (2 / 9, 1 / 3, 4 / 9, 0.0), (2 / 9 - 7 / 24, 1 / 3 - 1 / 4, 4 / 9 - 1 / 3, -1 / 8),
                             (0.0, 0.5, 0.0, 0.0)),
    # Fehlberg 2(1) (torchdiffeq fehlberg2.py)
    "fehlberg2": EmbeddedTableau("fehlberg2", 2, (1 / 2, 1.0), ((1 / 2,), (1 / 256, 255 / 256)),
                                 (1 / 512, 255 / 256, 1 / 512), (-1 / 512, 0.0, 1 / 512), (0.0, 0.5, 0.0)),
    # Heun-Euler 2(1) (torchdiffeq adaptive_heun.py)
    "adaptive_heun": EmbeddedTableau("adaptive_heun", 2, (1.0,), ((1.0,),), (0.5, 0.5), (0.5, -0.5), (0.5, 0.0)),
}


def _f32(v) -> torch.Tensor:
    return torch.as_tensor(v, dtype=torch.float32)


def _rms(t: torch.Tensor) -> torch.Tensor:
    return t.abs().pow(2).mean().sqrt()


def _mixed_norm(parts) -> float:
    """max over the components of the tuple state of their RMS norms (torchdiffeq `_mixed_norm`)."""
    return max(float(_rms(p)) for p in parts if p is not None and p.numel() > 0)


# A step function runs ONE kernel launch:
#   step(y, k1, lp0, kl1, t_rows [n] fp32 real times, cin [n,8], slots [n], tail_coef [4,8], use_y bits, n_aux)
#     -> (aux [n_aux,B,D], aux_lp [n_aux,B] or None)
StepFn = Callable[..., Tuple[torch.Tensor, Optional[torch.Tensor]]]


def _onehot(i):
    v = torch.zeros(8)
    v[i] = 1.0
    return v


class Dopri5:
    """Batch-global adaptive Dormand-Prince over an increasing solver-time span [t0, t1]."""

    def __init__(self, step: StepFn, has_lp: bool, rtol: float, atol: float, options: Optional[dict] = None,
                 norm_only=(), method: str = "dopri5"):
        """``norm_only``: components the reference carries in the tuple state with a zero derivative (the raw
        ``conditional`` of ConditionalODEFlow, flow.py:779-796, 855-881).  Under the mixed norm they can only
        matter where the state itself is measured -- d0 of the initial step; their derivative and error
        estimate are identically zero."""
        opts = dict(options or {})
        if method not in TABLEAUX:
            raise NotImplementedError(f"adaptive method {method!r}: supported {sorted(TABLEAUX)} (dopri8 needs 13 stage "
                                      "slots; the fused kernels keep 7 on chip)")
        self.tab = TABLEAUX[method]
        self.norm_only = [c for c in norm_only if c is not None and c.numel() > 0]
        self.step = step
        self.has_lp = has_lp
        self.rtol = float(rtol)
        self.atol = float(atol)
        self.min_step = float(opts.pop("min_step", 0.0))
        self.max_step = float(opts.pop("max_step", float("inf")))
        self.first_step = opts.pop("first_step", None)
        self.max_num_steps = int(opts.pop("max_num_steps", 2 ** 31 - 1))
        for k in ("step_t", "jump_t", "norm", "dtype"):
            if opts.get(k) is not None:
                raise NotImplementedError(f"dopri5 option {k!r} is not supported on the fused path")
        self.n_attempts = 0
        self.n_accepted = 0
        # the tableau as fp32 tensors, once per solve (an attempt then costs a handful of host tensor ops instead of ~40)
        S = self.tab.stages
        self._alpha = _f32(self.tab.alpha)
        self._alpha_is_one = torch.tensor([a == 1.0 for a in self.tab.alpha])
        self._beta8 = torch.zeros(S - 1, 8)
        for i, beta in enumerate(self.tab.beta):
            self._beta8[i, : len(beta)] = _f32(beta)
        pad = lambda c: torch.cat([_f32(c), torch.zeros(8 - S)])
        self._c_sol8, self._c_mid8, self._c_err8 = pad(self.tab.c_sol), pad(self.tab.c_mid), pad(self.tab.c_error)
        self._last_stage = _onehot(S - 1)
        self._stage_slots = torch.arange(1, S, dtype=torch.int32)

    # -- single launches ----------------------------------------------------------------------
    def _deriv(self, t, y, lp, k1=None, kl1=None, h=None):
        """f(t, y) if k1 is None, else f(t, y + h*k1) -- one evaluation row, slot 0 or 1."""
        slot = 0 if k1 is None else 1
        cin = torch.zeros(1, 8)
        if k1 is not None:
            cin[0, 0] = h
        aux, aux_lp = self.step(y, k1, lp, kl1, _f32([t]), cin, torch.tensor([slot], dtype=torch.int32),
                                torch.stack([_onehot(slot), torch.zeros(8), torch.zeros(8), torch.zeros(8)]), 0, 1)
        return aux[0], (aux_lp[0] if self.has_lp else None)

    def _attempt(self, t0, dt, t1, y, lp, f0, fl0):
        """Stages 2..7 of one step from (t0, y) with step dt; returns y1, lp1, f1, fl1, mids, errors."""
        t0f, dtf, t1f = _f32(t0), _f32(dt), _f32(t1)              # time enters the stages in the state dtype
        ts = torch.where(self._alpha_is_one, t1f, t0f + self._alpha * dtf)     # a stage at alpha = 1 sits at t1 itself
        cin = self._beta8 * dtf
        tail = torch.stack([dtf * self._c_sol8,                    # y1    = y + dt * k . c_sol
                            self._last_stage,                      # f1    = the last stage (FSAL)
                            dtf * self._c_mid8,                    # y_mid = y + dt * k . c_mid
                            dtf * self._c_err8])                   # err   = dt * k . c_error
        aux, aux_lp = self.step(y, f0, lp, fl0, ts, cin, self._stage_slots, tail, 0b0101, 4)
        if self.has_lp:
            return aux, aux_lp
        return aux, None

    # -- torchdiffeq's helpers, state in fp32 on the device, time in float64 on the host ----------
    def _norms(self, terms, check=None):
        """Mixed norm (max over the components of their scaled RMS norms) of `terms` = [(num, sub, scale0, scale1), ..],
        and whether `check` holds a non-finite value.  On the GPU: one ff_scaled_rms launch and one read-back, the
        only host synchronisation of an attempted step.  (CPU tensors -- the kernel-semantics emulator of the tests --
        take the same arithmetic in torch ops.)"""
        if terms[0][0].is_cuda:
            from . import _native
            out = []
            for i in range(0, len(terms), _native.NORM_TERMS):
                part = terms[i:i + _native.NORM_TERMS]
                vals = _native.scaled_rms([tuple(None if t is None else t.reshape(-1) for t in term) for term in part],
                                          self.atol, self.rtol, check.reshape(-1) if (check is not None and i == 0) else None)
                out.extend(vals[:len(part)])
                if i == 0:
                    bad = vals[len(part)] != 0.0
            return max(out), (bad if check is not None else False)
        parts = []
        for num, sub, s0, s1 in terms:
            sc = s0.abs() if s1 is None else torch.max(s0.abs(), s1.abs())
            parts.append((num if sub is None else num - sub) / (self.atol + self.rtol * sc))
        return _mixed_norm(parts), (not bool(torch.isfinite(check).all()) if check is not None else False)

    def _select_initial_step(self, t0, y, lp, f0, fl0):
        if y.is_cuda:
            ys = [p for p in (y, lp) if p is not None]
            fs = [p for p in (f0, fl0) if p is not None]
            d0, _ = self._norms([(a, None, a, None) for a in ys] + [(c, None, c, None) for c in self.norm_only])
            d1, _ = self._norms([(f, None, a, None) for f, a in zip(fs, ys)])
            h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
            h0 = float(_f32(abs(h0)))
            f1, fl1 = self._deriv(float(_f32(t0)) + h0, y, lp, k1=f0, kl1=fl0, h=h0)
            f1s = [p for p in (f1, fl1) if p is not None]
            d2 = abs(self._norms([(a, b, c, None) for a, b, c in zip(f1s, fs, ys)])[0] / h0)
            if d1 <= 1e-15 and d2 <= 1e-15:
                h1 = max(1e-6, h0 * 1e-3)
            else:
                h1 = (0.01 / max(d1, d2)) ** (1.0 / float(self.tab.order))
            return min(100 * h0, abs(h1))
        scale = [self.atol + p.abs() * self.rtol for p in (y, lp) if p is not None]
        ys = [p for p in (y, lp) if p is not None]
        fs = [p for p in (f0, fl0) if p is not None]
        d0 = _mixed_norm([a / s for a, s in zip(ys, scale)] +
                         [c / (self.atol + c.abs() * self.rtol) for c in self.norm_only])
        d1 = _mixed_norm([a / s for a, s in zip(fs, scale)])
        h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
        h0 = float(_f32(abs(h0)))
        f1, fl1 = self._deriv(float(_f32(t0)) + h0, y, lp, k1=f0, kl1=fl0, h=h0)
        f1s = [p for p in (f1, fl1) if p is not None]
        d2 = abs(_mixed_norm([(a - b) / s for a, b, s in zip(f1s, fs, scale)]) / h0)
        if d1 <= 1e-15 and d2 <= 1e-15:
            h1 = max(1e-6, h0 * 1e-3)
        else:
            h1 = (0.01 / max(d1, d2)) ** (1.0 / float(self.tab.order))      # called with order - 1 -> 1/((order-1)+1)
        return min(100 * h0, abs(h1))

    def _error_ratio(self, errs, y0s, y1s):
        parts = []
        for e, a, b in zip(errs, y0s, y1s):
            tol = self.atol + self.rtol * torch.max(a.abs(), b.abs())
            parts.append(e / tol)
        return _mixed_norm(parts)

    def _optimal_step_size(self, last_step, error_ratio):
        if error_ratio != error_ratio:          # NaN propagates (torch.min/max do), the next attempt raises
            return float("nan")
        if error_ratio == 0:
            return last_step * IFACTOR
        dfactor = 1.0 if error_ratio < 1 else DFACTOR
        factor = min(IFACTOR, max(SAFETY / error_ratio ** (1.0 / self.tab.order), dfactor))
        return last_step * factor

    def _fail(self, message: str) -> RuntimeError:
        """torchdiffeq's assertion as an exception that also says how far the solve got (``.solver_stats``)."""
        err = RuntimeError(message)
        err.solver_stats = {"attempts": self.n_attempts, "accepted": self.n_accepted}
        return err

    # -- driver ----------------------------------------------------------------------------------
    def integrate(self, t0: float, t_end: float, y: torch.Tensor, lp: Optional[torch.Tensor]):
        """Advance (y, lp) from solver time t0 to t_end > t0; returns the dense-output values there."""
        f0, fl0 = self._deriv(t0, y, lp)
        dt = self.first_step if self.first_step is not None else self._select_initial_step(t0, y, lp, f0, fl0)
        dt = float(dt)
        t_lo, t_hi = t0, t0                      # rk_state.t0, rk_state.t1
        interp = None
        n_steps = 0
        while t_end > t_hi:
            if n_steps >= self.max_num_steps:
                raise self._fail(f"max_num_steps exceeded ({n_steps}>={self.max_num_steps})")
            if dt == dt:
                dt = min(max(dt, self.min_step), self.max_step)      # every attempt starts from a clamped step
            ta, tb = t_hi, t_hi + dt
            if not (ta + dt > ta):      # also catches dt = NaN after a non-finite error estimate
                raise self._fail(f"underflow in dt {dt}")
            aux, aux_lp = self._attempt(ta, dt, tb, y, lp, f0, fl0)
            self.n_attempts += 1
            y1, f1, ymid, yerr = aux[0], aux[1], aux[2], aux[3]
            # error ratio (mixed norm of err / (atol + rtol max(|y0|, |y1|))) and the finiteness of the new state: one
            # reduction launch and one read-back on the GPU
            terms = [(yerr, None, y, y1)]
            if self.has_lp:
                lp1, fl1, lpmid, lperr = aux_lp[0], aux_lp[1], aux_lp[2], aux_lp[3]
                terms.append((lperr, None, lp, lp1))
            ratio, y1_bad = self._norms(terms, check=y1)
            accept = ratio <= 1
            if dt > self.max_step:
                accept = False
            if dt <= self.min_step:
                accept = True
            if accept:
                if y1_bad:
                    raise self._fail("non-finite values in state `y`")
                self.n_accepted += 1
                interp = (ta, tb, dt, y, y1, ymid, f0, f1,
                          (lp, lp1, lpmid, fl0, fl1) if self.has_lp else None)
                t_lo, t_hi = ta, tb
                y, f0 = y1, f1
                if self.has_lp:
                    lp, fl0 = lp1, fl1
            dt = self._optimal_step_size(dt, ratio)
            if dt == dt:
                dt = min(max(dt, self.min_step), self.max_step)
            n_steps += 1
        return self._interp(interp, t_end)

    @staticmethod
    def _fit_eval(y0, y1, y_mid, f0, f1, dt, x):
        dt = _f32(dt).to(y0.device)
        a = 2 * dt * (f1 - f0) - 8 * (y1 + y0) + 16 * y_mid
        b = dt * (5 * f0 - 3 * f1) + 18 * y0 + 14 * y1 - 32 * y_mid
        c = dt * (f1 - 4 * f0) - 11 * y0 - 5 * y1 + 16 * y_mid
        d = dt * f0
        e = y0
        x = _f32(x).to(y0.device)
        total = e + x * d
        xp = x
        for coef in (c, b, a):
            xp = xp * x
            total = total + xp * coef
        return total

    def _interp(self, interp, t):
        ta, tb, dt, y0, y1, ymid, f0, f1, lps = interp
        x = (t - ta) / (tb - ta)
        y = self._fit_eval(y0, y1, ymid, f0, f1, dt, x)
        lp = None
        if lps is not None:
            lp0, lp1, lpmid, fl0, fl1 = lps
            lp = self._fit_eval(lp0, lp1, lpmid, fl0, fl1, dt, x)
        return y, lp
