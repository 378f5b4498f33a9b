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

import math
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


# Dormand-Prince 8(7), 13 stages + the first-same-as-last stage (torchdiffeq dopri8.py: Prince & Dormand's RK8(7)13M).
# torchdiffeq is not available offline; the nodes, matrix, eighth-order weights and seventh-order embedded weights below are
# restated from memory of that file and VERIFIED, not trusted: row sums, every rooted-tree condition up to order 4 and the
# quadrature conditions sum b_i c_i^k = 1/(k+1) hold to 1e-15 for k <= 7 (b8) and k <= 6 (b7, which must and does fail
# k = 7 by 1e-4), and the eighth-order solution converges with order 8 (tests/test_oracle_known_answers.py).  A wrong digit
# anywhere breaks one of them at the 1e-10 level.
_DP8_C = (1 / 18, 1 / 12, 1 / 8, 5 / 16, 3 / 8, 59 / 400, 93 / 200, 5490023248 / 9719169821, 13 / 20,
          1201146811 / 1299019798, 1.0, 1.0)
_DP8_B8 = (14005451 / 335480064, 0, 0, 0, 0, -59238493 / 1068277825, 181606767 / 758867731, 561292985 / 797845732,
           -1041891430 / 1371343529, 760417239 / 1151165299, 118820643 / 751138087, -528747749 / 2220607170, 1 / 4)
_DP8_B7 = (13451932 / 455176623, 0, 0, 0, 0, -808719846 / 976000145, 1757004468 / 5645159321, 656045339 / 265891186,
           -3867574721 / 1518517206, 465885868 / 322736535, 53011238 / 667516719, 2 / 45, 0)
_DP8_A = (
    (1 / 18,),
    (1 / 48, 1 / 16),
    (1 / 32, 0, 3 / 32),
    (5 / 16, 0, -75 / 64, 75 / 64),
    (3 / 80, 0, 0, 3 / 16, 3 / 20),
    (29443841 / 614563906, 0, 0, 77736538 / 692538347, -28693883 / 1125000000, 23124283 / 1800000000),
    (16016141 / 946692911, 0, 0, 61564180 / 158732637, 22789713 / 633445777, 545815736 / 2771057229, -180193667 / 1043307555),
    (39632708 / 573591083, 0, 0, -433636366 / 683701615, -421739975 / 2616292301, 100302831 / 723423059,
     790204164 / 839813087, 800635310 / 3783071287),
    (246121993 / 1340847787, 0, 0, -37695042795 / 15268766246, -309121744 / 1061227803, -12992083 / 490766935,
     6005943493 / 2108947869, 393006217 / 1396673457, 123872331 / 1001029789),
    (-1028468189 / 846180014, 0, 0, 8478235783 / 508512852, 1311729495 / 1432422823, -10304129995 / 1701304382,
     -48777925059 / 3047939560, 15336726248 / 1032824649, -45442868181 / 3398467696, 3065993473 / 597172653),
    (185892177 / 718116043, 0, 0, -3185094517 / 667107341, -477755414 / 1098053517, -703635378 / 230739211,
     5731566787 / 1027545527, 5232866602 / 850066563, -4093664535 / 808688257, 3962137247 / 1805957418, 65686358 / 487910083),
    (403863854 / 491063109, 0, 0, -5068492393 / 434740067, -411421997 / 543043805, 652783627 / 914296604,
     11173962825 / 925320556, -13158990841 / 6184727034, 3936647629 / 1978049680, -160528059 / 685178525,
     248638103 / 1413531060, 0),
)
# Dense-output midpoint y(t0 + dt/2) ~ y0 + dt sum_i m_i k_i.  torchdiffeq evaluates a seventh-order continuous extension
# at 1/2 there (long decimal coefficients that cannot be recalled or re-derived digit for digit); these weights are this
# build's own: the minimum-norm solution, on the stages the eighth-order weights use, of ALL rooted-tree conditions up to
# order 5 at theta = 1/2 plus the quadrature conditions up to k = 7 (residual 2e-16; measured midpoint error ~ dt^6).
# torchdiffeq only uses the midpoint inside its quartic Hermite fit of the LAST step (error O(dt^5) whatever the midpoint's
# order), so the difference is far below any tolerance the solver is run at -- but it is a difference, and it is stated.
_DP8_MID = (0.04303473960045479, 0.0, 0.0, 0.0, 0.0, 0.10243459611345074, 0.22994756530151478, 0.2305207916012828,
            -0.17844101414677382, 0.07627042306944881, -0.0061595447067761405, 0.0007974810543300986,
            0.000797481057637911, 0.0007974810554298717)

# pairs with more stages than the fused kernels keep on chip (FF_MAX_SLOTS = 7): stepped stage by stage (HostSteppedPair)
WIDE_TABLEAUX = {
    "dopri8": EmbeddedTableau("dopri8", 8, _DP8_C + (1.0,), _DP8_A + (_DP8_B8,), _DP8_B8 + (0.0,),
                              tuple(a - b for a, b in zip(_DP8_B8, _DP8_B7)) + (0.0,), _DP8_MID),
}


def _f32(v) -> torch.Tensor:
    return torch.as_tensor(v, dtype=torch.float32)


def _rms(t: torch.Tensor) -> torch.Tensor:
    return t.abs().pow(2).mean().sqrt()


def _mixed_norm(parts) -> float:
    """max over the components of the tuple state of their RMS norms (torchdiffeq `_mixed_norm`).  Inside
    ``distributed.global_step_control`` the components are shards of a batch cut over ranks: their sums of squares and
    element counts meet before the root is taken."""
    from . import distributed
    exchange, group = distributed.step_control_group()
    if exchange:
        parts = [p for p in parts if p is not None]
        tot = distributed.sum_over_ranks([float(p.double().pow(2).sum()) for p in parts] + [float(p.numel()) for p in parts],
                                         parts[0].device, group)
        k = len(parts)
        return max(math.sqrt(tot[j] / tot[k + j]) for j in range(k) if tot[k + j] > 0)
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
                 norm_only=(), method: str = "dopri5", sign: float = 1.0):
        """``norm_only``: components the reference carries in the tuple state with a zero derivative (the raw
        ``conditional`` of ConditionalODEFlow, flow.py:779-796, 855-881).  Under the mixed norm they can only
        matter where the state itself is measured -- d0 of the initial step; their derivative and error
        estimate are identically zero.  ``sign`` = -1 for a decreasing span (solved in negated time): the options that
        name TIMES (``step_t``, ``jump_t``) are negated with it, as torchdiffeq's ``_check_inputs`` does."""
        if method not in TABLEAUX:
            raise NotImplementedError(f"adaptive method {method!r}: {sorted(TABLEAUX)} run one launch per attempted step; "
                                      f"{sorted(WIDE_TABLEAUX)} have more stages than the 7 slots the fused kernels keep on chip "
                                      "and take adaptive.HostSteppedPair (adaptive.make_solver picks)")
        self.tab = TABLEAUX[method]
        self.step = step
        self._init_control(has_lp, rtol, atol, options, norm_only, sign)
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

    def _init_control(self, has_lp, rtol, atol, options, norm_only, sign=1.0):
        """Tolerances and torchdiffeq's step-control options (shared by the one-launch-per-attempt and the stage-by-stage
        drivers).  ``step_t`` (times a step must END on), ``jump_t`` (the same, and the derivative is re-evaluated just
        behind them: discontinuities of the right-hand side) and ``norm`` (a callable on the tuple state replacing the
        mixed RMS norm) are what torchdiffeq's RKAdaptiveStepsizeODESolver takes besides the step-size limits; the host
        controller serves them (device_adaptive.supported sends such solves here)."""
        opts = dict(options or {})
        self.norm_only = [c for c in norm_only if c is not None and c.numel() > 0]
        self.has_lp = has_lp
        self.rtol = float(rtol)
        self.atol = float(atol)
        self.min_step = float(opts.pop("min_step", 0.0))
        self.max_step = float(opts.pop("max_step", float("inf")))
        self.first_step = opts.pop("first_step", None)
        self.max_num_steps = int(opts.pop("max_num_steps", 2 ** 31 - 1))
        if opts.get("dtype") not in (None, torch.float64):      # torchdiffeq's time dtype: float64 is its default and what runs here
            raise NotImplementedError("adaptive option dtype: time is kept in float64 (torchdiffeq's default); other dtypes are not built")
        def tvals(v):       # torchdiffeq `_sort_tvals` happens in integrate (it needs t0); here: float64, solver time
            return None if v is None else (float(sign) * torch.as_tensor(v, dtype=torch.float64).detach().reshape(-1).cpu())
        self.step_t, self.jump_t = tvals(opts.pop("step_t", None)), tvals(opts.pop("jump_t", None))
        self.norm_fn = opts.pop("norm", None)
        if self.norm_fn is not None and not callable(self.norm_fn):
            raise TypeError("options['norm'] must be a callable taking the tuple state and returning a scalar")
        self.n_attempts = 0
        self.n_accepted = 0

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
        if terms[0][0].is_cuda and self.norm_fn is None:
            from . import _native, distributed
            exchange, group = distributed.step_control_group()
            out = []
            for i in range(0, len(terms), _native.NORM_TERMS):
                part = terms[i:i + _native.NORM_TERMS]
                vals = _native.scaled_rms([tuple(None if t is None else t.reshape(-1) for t in term) for term in part],
                                          self.atol, self.rtol, check.reshape(-1) if (check is not None and i == 0) else None)
                if exchange:
                    # several shards of one batch (distributed.global_step_control): mean squares -> sums of squares,
                    # summed over the ranks with the element counts -> the norms of the whole batch
                    cnt = [float(term[0].numel()) for term in part]
                    tot = distributed.sum_over_ranks([float(v) ** 2 * n for v, n in zip(vals, cnt)] + cnt
                                                     + [float(vals[len(part)])], terms[0][0].device, group)
                    k = len(part)
                    vals = [math.sqrt(tot[j] / tot[k + j]) if tot[k + j] > 0 else 0.0 for j in range(k)] + [tot[2 * k]]
                out.extend(vals[:len(part)])
                if i == 0:
                    bad = vals[len(part)] != 0.0
            return max(out), (bad if check is not None else False)
        parts = []
        for num, sub, s0, s1 in terms:
            sc = s0.abs() if s1 is None else torch.max(s0.abs(), s1.abs())
            parts.append((num if sub is None else num - sub) / (self.atol + self.rtol * sc))
        bad = not bool(torch.isfinite(check).all()) if check is not None else False
        from . import distributed
        exchange, group = distributed.step_control_group()
        if exchange and check is not None:
            bad = distributed.sum_over_ranks([1.0 if bad else 0.0], check.device, group)[0] > 0
        return self._norm_of(parts), bad

    def _norm_of(self, parts, mids=None) -> float:
        """The norm of the scaled components ``parts`` = [y] or [y, lp]: torchdiffeq's mixed norm, or the user's
        ``options["norm"]`` on the tuple as the reference's state has it -- (x, delta_logp [B, 1]) for the score models
        (diffusion.py:744-752), (x[, conditional], logJ [B, 1]) for the flows (flow.py:371-382, 869-881).  ``mids``: the
        scaled values of the components carried with a zero derivative (they sit between the state and the divergence);
        None where a derivative or an error estimate is measured: zeros."""
        if self.norm_fn is None:
            return _mixed_norm(list(parts) + list(mids or []))
        from . import distributed
        if distributed.step_control_group()[0]:
            raise NotImplementedError("options['norm'] with distributed.global_step_control: a user norm cannot be summed over ranks")
        mids = list(mids) if mids is not None else [torch.zeros_like(c) for c in self.norm_only]
        tup = [parts[0]] + mids + ([parts[1].reshape(-1, 1)] if self.has_lp else [])
        return float(self.norm_fn(tuple(tup)))

    def _select_initial_step(self, t0, y, lp, f0, fl0):
        if y.is_cuda and self.norm_fn is None:
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
        cs = [c / (self.atol + c.abs() * self.rtol) for c in self.norm_only]
        scaled_y = [a / s for a, s in zip(ys, scale)]
        d0 = self._norm_of(scaled_y, cs)
        d1 = self._norm_of([a / s for a, s in zip(fs, scale)])
        h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
        h0 = float(_f32(abs(h0)))
        f1, fl1 = self._deriv(float(_f32(t0)) + h0, y, lp, k1=f0, kl1=fl0, h=h0)
        f1s = [p for p in (f1, fl1) if p is not None]
        d2 = abs(self._norm_of([(a - b) / s for a, b, s in zip(f1s, fs, scale)]) / h0)
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
        # options step_t / jump_t (torchdiffeq `_before_integrate`: `_sort_tvals` keeps the times >= t0, sorted; the two lists
        # must not share an element; the index of the next one by bisection)
        import bisect
        import numpy as np
        sort_t = lambda v: [] if v is None else sorted(float(u) for u in v.tolist() if float(u) >= t0)
        step_t, jump_t = sort_t(self.step_t), sort_t(self.jump_t)
        both = step_t + jump_t
        if len(set(both)) != len(both):
            raise ValueError("`step_t` and `jump_t` must not have any repeated elements between them.")
        i_step = min(bisect.bisect(step_t, t0), len(step_t) - 1)
        i_jump = min(bisect.bisect(jump_t, t0), len(jump_t) - 1)
        while t_end > t_hi:
            if n_steps >= self.max_num_steps:
                raise self._fail(f"max_num_steps exceeded ({n_steps}>={self.max_num_steps})")
            if dt == dt:
                dt = min(max(dt, self.min_step), self.max_step)      # every attempt starts from a clamped step
            ta, tb = t_hi, t_hi + dt
            if not (ta + dt > ta):      # also catches dt = NaN after a non-finite error estimate
                raise self._fail(f"underflow in dt {dt}")
            # a step that would cross the next step_t / jump_t ends ON it (`_adaptive_step`)
            on_step_t = on_jump_t = False
            if step_t:
                on_step_t = ta < step_t[i_step] < ta + dt
                if on_step_t:
                    tb = step_t[i_step]
                    dt = tb - ta
            if jump_t:
                on_jump_t = ta < jump_t[i_jump] < ta + dt
                if on_jump_t:
                    on_step_t = False
                    tb = jump_t[i_jump]
                    dt = tb - ta
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
                if on_step_t and i_step != len(step_t) - 1:
                    i_step += 1
                if on_jump_t:
                    if i_jump != len(jump_t) - 1:
                        i_jump += 1
                    # just past a discontinuity of the right-hand side: the derivative of the side we are on now --
                    # torchdiffeq evaluates `func(t_next, y_next, perturb=Perturb.NEXT)`, i.e. at the next representable
                    # time of the STATE's dtype (fp32) in solver time
                    t_after = float(np.nextafter(np.float32(tb), np.float32(np.inf)))
                    f0, fl0 = self._deriv(t_after, y, lp)
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


def _lincomb(x, ks, coefs, like, x_coef=1.0):
    """x_coef * x + sum_i coefs[i] * ks[i] over any number of terms: ff_stage_combine passes of up to 7 terms on the GPU
    (terms with a zero coefficient are never read); plain torch on CPU tensors (the tests' kernel-semantics emulator)."""
    terms = [(k, float(c)) for k, c in zip(ks, coefs) if k is not None and float(c) != 0.0]
    if not like.is_cuda:
        out = torch.zeros_like(like) if x is None or x_coef == 0.0 else x_coef * x
        for k, c in terms:
            out = out + c * k
        return out
    from . import _native
    out = torch.empty_like(like)
    if not terms:
        return _native.stage_combine(out, x, [], [], x_coef)
    first = True
    for i in range(0, len(terms), 7):
        part = terms[i:i + 7]
        _native.stage_combine(out, x if first else out, [k for k, _ in part], [c for _, c in part], x_coef if first else 1.0)
        first = False
    return out


class HostSteppedPair(Dopri5):
    """torchdiffeq's embedded pairs with more stages than the fused kernels keep on chip -- ``dopri8``: 13 stages and the
    first-same-as-last one against FF_MAX_SLOTS = 7 stage slots and 8 coefficient words per evaluation row.  The
    reference passes ``method=`` straight to ``odeint`` (diffusion.py:631-639, 744-752; flow.py:371-382), so the method is
    reachable; nothing in the reference uses it.  Same step control as ``Dopri5`` (inherited: initial step, mixed RMS
    norm, accept / reject, step-size law, dense output at the end); an attempted step is walked stage by stage on the
    host: every stage ONE right-hand-side evaluation -- a fused single-row launch of the network with its divergence, or
    the user's module -- and every stage input / solution / midpoint / error combination ``ff_stage_combine`` passes
    over the state.  13 + 1 evaluations per attempt whatever the driver, so at solver-sized batches the launches cost
    nothing against the evaluations; at notebook sizes it is launch-bound (~40 launches per attempt)."""

    def __init__(self, step: StepFn, has_lp: bool, rtol: float, atol: float, options: Optional[dict] = None,
                 norm_only=(), method: str = "dopri8", sign: float = 1.0):
        if method not in WIDE_TABLEAUX:
            raise NotImplementedError(f"stage-by-stage stepping is for {sorted(WIDE_TABLEAUX)}; {method!r} runs on Dopri5")
        self.tab = WIDE_TABLEAUX[method]
        self.step = step
        self._init_control(has_lp, rtol, atol, options, norm_only, sign)
        self._slot0 = torch.zeros(1, dtype=torch.int32)
        self._cin0 = torch.zeros(1, 8)
        self._tail0 = torch.stack([_onehot(0), torch.zeros(8), torch.zeros(8), torch.zeros(8)])

    def _rhs(self, t32: torch.Tensor, y, lp):
        """Solver-time derivative at fp32 time ``t32`` ([1] tensor): one evaluation row, slot 0, returned as aux_0."""
        aux, aux_lp = self.step(y, None, lp, None, t32, self._cin0, self._slot0, self._tail0, 0, 1)
        return aux[0], (aux_lp[0] if self.has_lp else None)

    def _deriv(self, t, y, lp, k1=None, kl1=None, h=None):
        yi = y if k1 is None else _lincomb(y, [k1], [float(_f32(h))], y)
        return self._rhs(_f32([t]), yi, lp)

    def _attempt(self, t0, dt, t1, y, lp, f0, fl0):
        tab = self.tab
        t0f, dtf, t1f = _f32(t0), _f32(dt), _f32(t1)              # time enters the stages in the state dtype
        ks, kl = [f0], [fl0]
        for alpha, beta in zip(tab.alpha, tab.beta):
            ti = t1f if alpha == 1.0 else t0f + _f32(alpha) * dtf
            yi = _lincomb(y, ks, [float(_f32(b) * dtf) for b in beta], y)
            f, fl = self._rhs(ti.reshape(1), yi, lp)
            ks.append(f)
            kl.append(fl)
        scaled = lambda c: [float(_f32(v) * dtf) for v in c]
        aux = [_lincomb(y, ks, scaled(tab.c_sol), y), ks[-1], _lincomb(y, ks, scaled(tab.c_mid), y),
               _lincomb(None, ks, scaled(tab.c_error), y, 0.0)]
        aux_lp = None
        if self.has_lp:
            aux_lp = [_lincomb(lp, kl, scaled(tab.c_sol), lp), kl[-1], _lincomb(lp, kl, scaled(tab.c_mid), lp),
                      _lincomb(None, kl, scaled(tab.c_error), lp, 0.0)]
        return aux, aux_lp


def make_solver(step: StepFn, has_lp: bool, rtol: float, atol: float, options: Optional[dict] = None, norm_only=(),
                method: str = "dopri5", sign: float = 1.0) -> Dopri5:
    """The adaptive driver for ``method``: one launch per attempted step (``Dopri5``) for the pairs whose stages fit the
    fused kernels' slots, stage by stage (``HostSteppedPair``) for ``dopri8``.  ``step`` has the contract of
    ``FusedNet.make_step`` / ``generic.ModuleStepper.make_step`` / ``host_stepper.RowStepper.make_step`` either way."""
    if method in WIDE_TABLEAUX:
        return HostSteppedPair(step, has_lp, rtol, atol, options, norm_only=norm_only, method=method, sign=sign)
    return Dopri5(step, has_lp, rtol, atol, options, norm_only=norm_only, method=method, sign=sign)
