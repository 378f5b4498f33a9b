"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain-torch, CPU restatement of the reference algorithm for the hot path (flowfusion's
probability-flow ODE / reverse-SDE sampling and log-density evaluation).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file; the
product (``flowfusion_amd/``) never does.

It deliberately keeps the reference's *unfused* op sequence (embed -> concat -> Linear/SiLU stack ->
score scaling -> drift, autograd for the divergence, a Python stepping loop), so it doubles as the
"port" CPU baseline.  Every function cites the reference lines it follows
(``diffusion.py`` = /root/reference/flowfusion/diffusion.py, ``flow.py`` likewise).

Pinning:
  * network / SDE schedules / RHS / divergence / Euler-Maruyama loop: pinned against golden vectors
    produced by the reference's own code (tests/golden/*.npz, made by tests/golden/make_golden.py).
  * ODE time stepping: the reference delegates it to torchdiffeq (>=0.2.5,<0.3.0,
    pyproject.toml:12), which is NOT in /root/reference and not installed offline.  ``odeint_fixed``
    restates torchdiffeq's published fixed-grid algorithm from memory of that package --
    PARITY UNPINNED for the stepper itself.  It is anchored by (a) hybrid vectors: this stepper
    driving the reference's own RHS (tests/golden), (b) known-answer tests with analytic scores
    (convergence order), see tests/test_oracle_*.py, (c) scipy (in this image): its RK45 / RK23 hold the
    same tableaux (numbers equal to 1e-16) and the same Hairer step control -- given this file's error
    weights, scipy's stepper and ``odeint_dopri5`` take the same steps (tests/test_oracle_known_answers.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch


# =================================================================================================
# parameters
# =================================================================================================
@dataclass
class MLPParams:
    """Plain tensors of the reference's score network (state_dict keys model.W, model.pi,
    model.NN.{i}.weight/bias)."""
    W: torch.Tensor                 # [E/2] embedding frequencies
    pi: torch.Tensor                # 0-dim
    weights: List[torch.Tensor]     # [out, in] per Linear
    biases: List[torch.Tensor]

    def to(self, dtype):
        return MLPParams(self.W.to(dtype), self.pi.to(dtype), [w.to(dtype) for w in self.weights],
                         [b.to(dtype) for b in self.biases])


def mlp_params_from_state_dict(sd, prefix="model.") -> MLPParams:
    n = 0
    while f"{prefix}NN.{n}.weight" in sd:
        n += 1
    return MLPParams(W=sd[f"{prefix}W"].detach().clone(), pi=sd[f"{prefix}pi"].detach().clone(),
                     weights=[sd[f"{prefix}NN.{i}.weight"].detach().clone() for i in range(n)],
                     biases=[sd[f"{prefix}NN.{i}.bias"].detach().clone() for i in range(n)])


def silu(x):
    return x * torch.sigmoid(x)


def mlp_forward(p: MLPParams, t, x, conditional=None, activation=silu):
    """diffusion.py:82-121 -- concat conditional, broadcast scalar t, Gaussian-Fourier features
    ``t*W*2*pi`` (in that order), ``[sin, cos, x]``, Linear/activation stack, final Linear.
    ``activation`` is the constructor's ``activation`` module (diffusion.py:38,77; default SiLU)."""
    if conditional is not None:
        x = torch.cat([x, conditional], dim=1)                      # :101-102
    if t.dim() == 0:
        t = t * torch.ones(x.shape[:-1], dtype=x.dtype)             # :105-106
    t_proj = t[:, None] * p.W[None, :] * 2 * p.pi                   # :109
    h = torch.cat([torch.sin(t_proj), torch.cos(t_proj), x], dim=1)  # :110-113
    for w, b in zip(p.weights[:-1], p.biases[:-1]):                  # :116-118
        h = activation(torch.nn.functional.linear(h, w, b))
    return torch.nn.functional.linear(h, p.weights[-1], p.biases[-1])   # :119


# =================================================================================================
# SDE schedules
# =================================================================================================
class VP:
    """diffusion.py:1006-1180."""
    kind = "vp"

    def __init__(self, beta_min=0.1, beta_max=20, T=1.0, epsilon=1e-3, dtype=torch.float32):
        self.beta_min, self.beta_max, self.T = beta_min, beta_max, T
        self.epsilon = torch.tensor(epsilon, dtype=torch.float32).to(dtype)
        self.dtype = dtype

    def beta(self, t):                                               # :1061
        return self.beta_min + (self.beta_max - self.beta_min) * (t / self.T)

    def marginal_prob_scalars(self, t):                              # :1152-1156
        log_coeff = 0.5 * (self.beta_max - self.beta_min) * t ** 2 / self.T + self.beta_min * t
        return torch.exp(-0.5 * log_coeff), torch.sqrt(1.0 - torch.exp(-log_coeff))

    def sigma(self, t):                                              # :1077
        return self.marginal_prob_scalars(t)[1]

    def diffusion(self, t, x):                                       # :1111-1112
        return torch.sqrt(self.beta(t)).view(-1, *[1] * (x.dim() - 1))

    def drift(self, t, x):                                           # :1130-1131
        return -0.5 * self.beta(t).view(-1, *[1] * (x.dim() - 1)) * x

    def prior_scale(self):                                           # :1093  Normal(0, 1)
        return torch.tensor(1.0, dtype=self.dtype)

    base_scale = None                                                # ODE base is not rescaled (:607-608)

    def T_value(self):
        return torch.tensor(self.T, dtype=self.dtype)


class SubVP(VP):
    """diffusion.py:1183-1366."""
    kind = "subvp"

    def diffusion(self, t, x):                                       # :1287-1297
        return torch.sqrt(
            self.beta(t) * (1.0 - torch.exp(-2 * self.beta_min * t - (self.beta_max - self.beta_min) * t ** 2 / self.T))
        ).view(-1, *[1] * (x.dim() - 1))

    def marginal_prob_scalars(self, t):                              # :1337-1342
        log_coeff = 0.5 * (self.beta_max - self.beta_min) * t ** 2 / self.T + self.beta_min * t
        return torch.exp(-0.5 * log_coeff), 1.0 - torch.exp(-log_coeff)


class VE:
    """diffusion.py:818-1003 (all four scalars are fp32 buffers there)."""
    kind = "ve"

    def __init__(self, sigma_min=1e-2, sigma_max=10.0, T=1.0, epsilon=1e-5, dtype=torch.float32):
        f = lambda v: torch.tensor(v, dtype=torch.float32).to(dtype)
        self.T, self.epsilon, self.sigma_min, self.sigma_max = f(T), f(epsilon), f(sigma_min), f(sigma_max)
        self.dtype = dtype

    def sigma(self, t):                                              # :866
        return self.sigma_min * (self.sigma_max / self.sigma_min) ** (t / self.T)

    def diffusion(self, t, x):                                       # :884-887
        return self.sigma(t).view(-1, *[1] * (x.dim() - 1)) * torch.sqrt(
            2 * (torch.log(self.sigma_max) - torch.log(self.sigma_min)) / self.T)

    def drift(self, t, x):                                           # :905
        return torch.zeros_like(x)

    def prior_scale(self):                                           # :1003  Normal(0, sigma_max)
        return self.sigma_max

    @property
    def base_scale(self):                                            # :605-606
        return self.sigma_max

    def T_value(self):
        return self.T


def normal_log_prob(x, scale):
    """torch.distributions.Normal(0, scale).log_prob(x) written out (used at diffusion.py:814)."""
    scale = torch.as_tensor(scale, dtype=x.dtype)
    var = scale ** 2
    return -(x ** 2) / (2 * var) - torch.log(scale) - math.log(math.sqrt(2 * math.pi))


# =================================================================================================
# score model pieces
def _hutchpp(A, S, G):
    """diffusion.py:336-399.  A(V): probes [n,B,D] -> J^T V as columns [B,D,n]."""
    Y = A(S)                                                          # :374-376
    Q, _ = torch.linalg.qr(Y, mode="reduced")                         # :379
    AQ = A(Q.permute(2, 0, 1))                                        # :383-386
    low_rank = torch.einsum("bdk,bdk->b", Q, AQ)                      # :387
    Gp = G.permute(1, 2, 0)                                           # :390
    U = Gp - torch.einsum("bdk,bkm->bdm", Q, torch.einsum("bdk,bdm->bkm", Q, Gp))   # :391-393
    AU = A(U.permute(2, 0, 1))                                        # :395-397
    return low_rank + torch.einsum("bdm,bdm->b", U, AU) / float(G.shape[0])         # :398-400


def _xtrace(A, O):
    """diffusion.py:401-481 (leave-one-out Hutch++ from a single QR)."""
    Y = A(O)                                                          # :446-448
    Q, R = torch.linalg.qr(Y, mode="reduced")                         # :451
    k = Q.shape[2]
    Z = A(Q.permute(2, 0, 1))                                         # :456-459
    H = torch.einsum("bdi,bdj->bij", Q, Z)                            # :461
    W = torch.einsum("bdk,mbd->bkm", Q, O)                            # :463
    T = torch.einsum("bdk,mbd->bkm", Z, O)                            # :465
    St = torch.linalg.solve_triangular(R, torch.eye(k, dtype=R.dtype), upper=True)     # :467
    St = St / torch.linalg.vector_norm(St, dim=-1, keepdim=True)      # :469
    S = St.permute(0, 2, 1)                                           # :470
    X = W - torch.sum(S * W, dim=1, keepdim=True) * S                 # :477
    per_probe = (torch.diagonal(H, 0, 1, 2).sum(-1)[:, None]          # :473, :489
                 - torch.sum(S * torch.einsum("bim,bmk->bik", H, S), dim=1)
                 + torch.sum(W * S, dim=1) * torch.sum(S * R, dim=1)
                 - torch.sum(T * X, dim=1)
                 + torch.sum(X * torch.einsum("bim,bmk->bik", H, X), dim=1))
    return per_probe.mean(dim=1)                                      # :491


# =================================================================================================
class ScoreOracle:
    """diffusion.py:124-815 restated over plain tensors."""

    def __init__(self, params: MLPParams, sde, no_sigma=False, dtype=torch.float32, activation=silu):
        self.p = params.to(dtype)
        self.sde = sde
        self.no_sigma = no_sigma
        self.dtype = dtype
        self.activation = activation                                 # MLP(activation=...), diffusion.py:38

    def score(self, t, x, conditional=None):                         # :233-238
        out = mlp_forward(self.p, t, x, conditional, self.activation)
        if self.no_sigma:
            return out
        return out / self.sde.sigma(t).view(-1, *[1] * len(x.shape[1:]))

    def ode_drift(self, t, x, conditional=None):                     # :276-279
        f = self.sde.drift(t, x)
        g = self.sde.diffusion(t, x)
        return f - 0.5 * g ** 2 * self.score(t, x, conditional=conditional)

    def rhs(self, t, states, conditional=None, divergence: Optional[str] = None, e=None):
        """diffusion.py:281-334, 483-508: ``states=(x,)`` -> xdot ; with ``divergence`` in
        {"hutch","exact","hutchpp","xtrace"} ``states=(x, dlogp)`` -> (xdot, div[B,1]).  ``e`` is the
        Hutchinson probe [B,D], the pair (S [r,B,D], G [m,B,D]) for Hutch++, or O [m,B,D] for XTrace."""
        x = states[0]
        if divergence is None:
            with torch.no_grad():
                return (self.ode_drift(t, x, conditional),)
        with torch.enable_grad():
            x = x.detach().requires_grad_(True)                      # :319
            x_dot = self.ode_drift(t, x, conditional)                # :323
            if divergence == "hutch":                                # :329-334
                div = torch.sum(torch.autograd.grad(x_dot, x, e, retain_graph=False)[0] * e, dim=1)
            elif divergence == "exact":                              # :483-503 (trace of the Jacobian)
                div = torch.zeros(x.shape[0], dtype=x.dtype)
                for i in range(x.shape[1]):
                    div = div + torch.autograd.grad(x_dot[:, i].sum(), x, retain_graph=True)[0][:, i]
            elif divergence in ("hutchpp", "xtrace"):
                # reverse-mode products A V = J^T V, one autograd call per probe (the reference batches
                # them with vmap, :370-372); probes [n,B,D] in, columns [B,D,n] out
                def A(V):
                    return torch.stack([torch.autograd.grad(x_dot, x, v, retain_graph=True)[0] for v in V], dim=2).detach()
                div = _hutchpp(A, *e) if divergence == "hutchpp" else _xtrace(A, e)
            else:
                raise ValueError(divergence)
        return x_dot.detach(), div.detach().view(x.shape[0], 1)       # :505-506

    # ---- solves ------------------------------------------------------------------------------------
    def sample_ode_from_base(self, base, conditional=None, method="rk4", options=None, atol=1e-4, rtol=1e-4):
        """diffusion.py:566-640 (returns just the samples)."""
        z = base * self.sde.base_scale if self.sde.base_scale is not None else base     # :605-608
        times = torch.stack([torch.tensor(1.0, dtype=torch.float32), self.sde.epsilon.to(torch.float32)]).to(self.dtype)  # :611
        func = lambda t, y: self.rhs(t, y, conditional, None)
        (traj,) = odeint(func, (z,), times, method, options, rtol, atol)
        return traj

    def solve_odes_forward(self, x0, conditional=None, method="rk4", options=None, divergence="hutch", e=None,
                           atol=1e-5, rtol=1e-5):
        """diffusion.py:642-754 -> (xT, delta_logp[B,1])."""
        dlogp = torch.zeros(x0.shape[0], 1, dtype=x0.dtype)         # :724
        times = torch.stack([self.sde.epsilon.to(torch.float32), torch.tensor(1.0, dtype=torch.float32)]).to(self.dtype)  # :727
        func = lambda t, y: self.rhs(t, y, conditional, divergence, e)
        return odeint(func, (x0, dlogp), times, method, options, rtol, atol)

    def log_prob(self, x0, conditional=None, method="rk4", options=None, divergence="hutch", e=None,
                 atol=1e-4, rtol=1e-4):
        """diffusion.py:756-815 -> [B,1]."""
        xT, lp = self.solve_odes_forward(x0, conditional, method, options, divergence, e, atol, rtol)
        return lp + torch.sum(normal_log_prob(xT, self.sde.prior_scale()), dim=1, keepdim=True)   # :814

    def sample_sde(self, x_prior, noise: Sequence[torch.Tensor], conditional=None, steps=100):
        """diffusion.py:510-563 with the random draws supplied by the caller: ``x_prior`` is the
        prior sample (:532-536), ``noise[i]`` the i-th ``randn_like`` (:554).  Returns x_mean (:563)."""
        x = x_prior
        batch = x.shape[0]
        dt = -(self.sde.T_value() - self.sde.epsilon) / steps         # :539
        t = torch.ones(batch, dtype=self.dtype) * self.sde.T_value()  # :540
        x_mean = x
        with torch.no_grad():
            for i in range(steps):                                    # :543
                if t[0] < self.sde.epsilon:                           # :548-551
                    break
                g = self.sde.diffusion(t, x)                          # :552
                f = self.sde.drift(t, x) - g ** 2 * self.score(t, x, conditional=conditional)   # :553
                dw = noise[i] * (-dt) ** (1.0 / 2.0)                  # :554-556
                x_mean = x + f * dt                                   # :557
                x = x_mean + g * dw                                   # :558
                t = t + dt                                            # :559
                if torch.any(torch.isnan(x)):                         # :560-562
                    break
        return x_mean


class PopulationOracle:
    """diffusion.py:1466-1640 (PopulationModelDiffusion) and :1643-1848 (...Conditional) restated: affine
    pre-/post-processing around a ScoreOracle.  ``cshift``/``cscale`` None = the unconditional wrapper."""

    def __init__(self, score: ScoreOracle, shift, scale, cshift=None, cscale=None, method="dopri5", options=None):
        dt = score.dtype
        self.score, self.method, self.options = score, method, options
        self.shift, self.scale = shift.to(dt), scale.to(dt)
        self.cshift = None if cshift is None else cshift.to(dt)
        self.cscale = None if cscale is None else cscale.to(dt)

    def _cond(self, conditional):                                     # :1776, :1809, :1839
        return None if self.cshift is None else (conditional - self.cshift) / self.cscale

    def forward(self, base, conditional=None):
        """:1575-1585 / :1772-1784: sample_ode_from_base(method=self.method, atol=rtol=1e-5) * scale + shift."""
        x = self.score.sample_ode_from_base(base, self._cond(conditional), self.method, self.options, 1e-5, 1e-5)
        return x * self.scale + self.shift

    def sample_sde(self, x_prior, noise, conditional=None):
        """:1608 / :1805-1814: always 100 steps (the wrapper ignores its `steps`), then * scale + shift."""
        return self.score.sample_sde(x_prior, noise, self._cond(conditional), steps=100) * self.scale + self.shift

    def log_prob(self, x, conditional=None, atol=1e-5, rtol=1e-5, divergence="exact", e=None):
        """:1632-1640 / :1837-1848: solve_odes_forward((x - shift)/scale, atol, rtol, options=self.options) --
        `self.method` is NOT forwarded, so the solver default (dopri5) applies -- plus the prior log-density;
        no log|scale| term."""
        xT, lp = self.score.solve_odes_forward((x - self.shift) / self.scale, self._cond(conditional), "dopri5",
                                               self.options, divergence, e, atol, rtol)
        return lp + torch.sum(normal_log_prob(xT, self.score.sde.prior_scale()), dim=1, keepdim=True)


# =================================================================================================
# fixed-grid ODE stepping (restatement of torchdiffeq's algorithm -- see the module docstring)
# =================================================================================================
_ONE_THIRD = 1.0 / 3.0
_TWO_THIRDS = 2.0 / 3.0


def _axpy(y, a, k):
    return tuple(yi + a * ki for yi, ki in zip(y, k))


def _step_euler(func, t0, dt, t1, y0):
    f0 = func(t0, y0)
    return tuple(dt * f for f in f0)


def _step_midpoint(func, t0, dt, t1, y0):
    half_dt = 0.5 * dt
    f0 = func(t0, y0)
    y_mid = _axpy(y0, half_dt, f0)
    return tuple(dt * f for f in func(t0 + half_dt, y_mid))


def _step_heun3(func, t0, dt, t1, y0):
    k1 = func(t0, y0)
    k2 = func(t0 + dt * _ONE_THIRD, tuple(y + dt * a * _ONE_THIRD for y, a in zip(y0, k1)))
    k3 = func(t0 + dt * _TWO_THIRDS, tuple(y + dt * (a * 0.0 + b * _TWO_THIRDS) for y, a, b in zip(y0, k1, k2)))
    return tuple(dt * (a * 0.25 + b * 0.0 + c * 0.75) for a, b, c in zip(k1, k2, k3))


def _step_rk4_38(func, t0, dt, t1, y0):
    """torchdiffeq's ``rk4`` = 3/8 rule ("rk4_alt_step_func")."""
    k1 = func(t0, y0)
    k2 = func(t0 + dt * _ONE_THIRD, tuple(y + dt * a * _ONE_THIRD for y, a in zip(y0, k1)))
    k3 = func(t0 + dt * _TWO_THIRDS, tuple(y + dt * (b - a * _ONE_THIRD) for y, a, b in zip(y0, k1, k2)))
    k4 = func(t1, tuple(y + dt * (a - b + c) for y, a, b, c in zip(y0, k1, k2, k3)))
    return tuple((a + 3 * (b + c) + d) * dt * 0.125 for a, b, c, d in zip(k1, k2, k3, k4))


def _step_rk4_classic(func, t0, dt, t1, y0):
    half_dt = dt * 0.5
    k1 = func(t0, y0)
    k2 = func(t0 + half_dt, _axpy(y0, half_dt, k1))
    k3 = func(t0 + half_dt, _axpy(y0, half_dt, k2))
    k4 = func(t1, _axpy(y0, dt, k3))
    return tuple((a + 2 * (b + c) + d) * dt * (1.0 / 6.0) for a, b, c, d in zip(k1, k2, k3, k4))


_DP_C = (0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0)
_DP_A = ((), (1 / 5,), (3 / 40, 9 / 40), (44 / 45, -56 / 15, 32 / 9),
         (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
         (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656))
_DP_B = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84)


def _step_dopri5_fixed(func, t0, dt, t1, y0):
    """Dormand-Prince 5th-order weights on a fixed step (extension; not a torchdiffeq method name)."""
    ks = []
    for i in range(6):
        yi = y0
        for j, a in enumerate(_DP_A[i]):
            yi = _axpy(yi, dt * a, ks[j])
        ti = t1 if i == 5 else (t0 if i == 0 else t0 + dt * _DP_C[i])
        ks.append(func(ti, yi))
    out = tuple(torch.zeros_like(y) for y in y0)
    for j, b in enumerate(_DP_B):
        out = _axpy(out, dt * b, ks[j])
    return out


_STEPPERS = {"euler": _step_euler, "midpoint": _step_midpoint, "heun3": _step_heun3, "rk4": _step_rk4_38,
             "rk4_classic": _step_rk4_classic, "dopri5_fixed": _step_dopri5_fixed}


def grid_from_step_size(t, step_size):
    start_time, end_time = t[0], t[-1]
    niters = torch.ceil((end_time - start_time) / step_size + 1).item()
    t_infer = torch.arange(0, niters, dtype=t.dtype) * step_size + start_time
    t_infer[-1] = t[-1]
    return t_infer


def odeint_fixed(func: Callable, y0: Tuple[torch.Tensor, ...], t: torch.Tensor, method="rk4", options=None):
    """Integrate ``dy/dt = func(t, y)`` from ``t[0]`` to ``t[-1]`` on a fixed grid; returns the tuple
    of states at ``t[-1]``.

    Restates torchdiffeq.odeint for ``method`` in euler/midpoint/heun3/rk4: decreasing ``t`` is solved
    as ``-t`` with ``func`` negated; with ``options["step_size"]`` the grid is
    ``arange(ceil((t1-t0)/h + 1))*h + t0`` with its last point set to ``t1``, otherwise the grid is
    ``t`` itself (a single step); tuple states advance component-wise (torchdiffeq flattens them
    into one vector, which is the same arithmetic).
    """
    options = dict(options or {})
    step_size = options.get("step_size", None)
    step = _STEPPERS[method]
    grid_constructor = options.get("grid_constructor")
    if bool(t[0] > t[-1]):
        t = -t
        base = func
        func = lambda tt, yy: tuple(-f for f in base(-tt, yy))
        if grid_constructor is not None:
            # torchdiffeq `_check_inputs`: options['grid_constructor'] = lambda func, y0, t: -_grid_constructor(func, y0, -t)
            # -- the user's constructor sees and returns real (decreasing) times
            user_gc = grid_constructor
            grid_constructor = lambda f, y, tt: -user_gc(f, y, -tt)
    # FixedGridODESolver's other options: grid_constructor(func, y0, t) in place of step_size, and
    # perturb: the step's first evaluation at nextafter(t0, +inf), an evaluation taken at t1 itself at nextafter(t1, -inf)
    # (_PerturbFunc wraps the already reversed function, so the shift is applied in solver time)
    if grid_constructor is not None:
        assert step_size is None, "step_size and grid_constructor are mutually exclusive arguments."
        grid = grid_constructor(func, y0, t)
        assert grid[0] == t[0] and grid[-1] == t[-1]
    else:
        grid = t if step_size is None else grid_from_step_size(t, step_size)
    y = tuple(y0)
    for t0, t1 in zip(grid[:-1], grid[1:]):
        dt = t1 - t0
        f = func
        if options.get("perturb"):
            # the steppers hand `t0` / `t1` themselves to func for those two evaluations, computed times otherwise
            f = lambda tt, yy, t0=t0, t1=t1: func(torch.nextafter(tt, tt + 1) if tt is t0 else
                                                  (torch.nextafter(tt, tt - 1) if tt is t1 else tt), yy)
        dy = step(f, t0, dt, t1, y)
        y = tuple(a + b for a, b in zip(y, dy))
    return y


# =================================================================================================
# adaptive Dormand-Prince 5(4)  (restatement of torchdiffeq's dopri5 -- parity unpinned, see header)
# =================================================================================================
_DP5_ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
_DP5_BETA = [[1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9],
             [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
             [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
             [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]]
_DP5_C_SOL = [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0]
_DP5_C_ERR = [35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
              -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1.0 / 60.0]
_DP5_C_MID = [6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
              187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


def _tuple_norm(parts):
    """torchdiffeq's default norm for tuple states: max over components of the RMS norm."""
    return max(p.abs().pow(2).mean().sqrt() for p in parts)


# the other embedded pairs torchdiffeq offers with at most 7 stages (bosh3.py, fehlberg2.py, adaptive_heun.py):
# name -> (order, alpha, beta, c_sol, c_error, c_mid)
_ADAPTIVE_TABLEAUX = {
    "bosh3": (3, [1 / 2, 3 / 4, 1.0], [[1 / 2], [0.0, 3 / 4], [2 / 9, 1 / 3, 4 / 9]], [2 / 9, 1 / 3, 4 / 9, 0.0],
              [2 / 9 - 7 / 24, 1 / 3 - 1 / 4, 4 / 9 - 1 / 3, -1 / 8], [0.0, 0.5, 0.0, 0.0]),
    "fehlberg2": (2, [1 / 2, 1.0], [[1 / 2], [1 / 256, 255 / 256]], [1 / 512, 255 / 256, 1 / 512],
                  [-1 / 512, 0.0, 1 / 512], [0.0, 0.5, 0.0]),
    "adaptive_heun": (2, [1.0], [[1.0]], [0.5, 0.5], [0.5, -0.5], [0.5, 0.0]),
}


# Dormand-Prince 8(7) (torchdiffeq dopri8.py: Prince & Dormand RK8(7)13M, 13 stages + the first-same-as-last one).  Restated
# from memory like the rest of the stepper -- but checkable, and checked (tests/test_oracle_known_answers.py): order conditions of
# the eighth-order weights b8 and of the embedded seventh-order weights b7, row sums, observed convergence order 8.  The
# dense-output midpoint weights are this restatement's own (torchdiffeq's come from a continuous extension whose decimal
# coefficients are not available offline): they satisfy every order condition up to 5 at theta = 1/2.
_DP8_NODES = [1 / 18, 1 / 12, 1 / 8, 5 / 16, 3 / 8, 59 / 400, 93 / 200, 5490023248 / 9719169821, 13 / 20,
              1201146811 / 1299019798, 1.0, 1.0]
_DP8_MATRIX = [
    [1 / 18],
    [1 / 48, 1 / 16],
    [1 / 32, 0, 3 / 32],
    [5 / 16, 0, -75 / 64, 75 / 64],
    [3 / 80, 0, 0, 3 / 16, 3 / 20],
    [29443841 / 614563906, 0, 0, 77736538 / 692538347, -28693883 / 1125000000, 23124283 / 1800000000],
    [16016141 / 946692911, 0, 0, 61564180 / 158732637, 22789713 / 633445777, 545815736 / 2771057229,
     -180193667 / 1043307555],
    [39632708 / 573591083, 0, 0, -433636366 / 683701615, -421739975 / 2616292301, 100302831 / 723423059,
     790204164 / 839813087, 800635310 / 3783071287],
    [246121993 / 1340847787, 0, 0, -37695042795 / 15268766246, -309121744 / 1061227803, -12992083 / 490766935,
     6005943493 / 2108947869, 393006217 / 1396673457, 123872331 / 1001029789],
    [-1028468189 / 846180014, 0, 0, 8478235783 / 508512852, 1311729495 / 1432422823, -10304129995 / 1701304382,
     -48777925059 / 3047939560, 15336726248 / 1032824649, -45442868181 / 3398467696, 3065993473 / 597172653],
    [185892177 / 718116043, 0, 0, -3185094517 / 667107341, -477755414 / 1098053517, -703635378 / 230739211,
     5731566787 / 1027545527, 5232866602 / 850066563, -4093664535 / 808688257, 3962137247 / 1805957418,
     65686358 / 487910083],
    [403863854 / 491063109, 0, 0, -5068492393 / 434740067, -411421997 / 543043805, 652783627 / 914296604,
     11173962825 / 925320556, -13158990841 / 6184727034, 3936647629 / 1978049680, -160528059 / 685178525,
     248638103 / 1413531060, 0],
]
_DP8_W8 = [14005451 / 335480064, 0, 0, 0, 0, -59238493 / 1068277825, 181606767 / 758867731, 561292985 / 797845732,
           -1041891430 / 1371343529, 760417239 / 1151165299, 118820643 / 751138087, -528747749 / 2220607170, 1 / 4]
_DP8_W7 = [13451932 / 455176623, 0, 0, 0, 0, -808719846 / 976000145, 1757004468 / 5645159321, 656045339 / 265891186,
           -3867574721 / 1518517206, 465885868 / 322736535, 53011238 / 667516719, 2 / 45, 0]
_DP8_MIDPOINT = [0.04303473960045479, 0.0, 0.0, 0.0, 0.0, 0.10243459611345074, 0.22994756530151478, 0.2305207916012828,
                 -0.17844101414677382, 0.07627042306944881, -0.0061595447067761405, 0.0007974810543300986,
                 0.000797481057637911, 0.0007974810554298717]
_ADAPTIVE_TABLEAUX["dopri8"] = (8, _DP8_NODES + [1.0], _DP8_MATRIX + [_DP8_W8], _DP8_W8 + [0.0],
                                [a - b for a, b in zip(_DP8_W8, _DP8_W7)] + [0.0], _DP8_MIDPOINT)


# attempts / accepted steps of the latest adaptive solve (kept up to date while it runs, so a solve that ends in
# torchdiffeq's "underflow in dt" assertion still says how far it got)
last_adaptive_stats = {"attempts": 0, "accepted": 0}


def odeint_dopri5(func, y0, t, rtol=1e-7, atol=1e-9, options=None, method="dopri5"):
    """Adaptive dopri5 (or another embedded pair of `_ADAPTIVE_TABLEAUX`) from ``t[0]`` to ``t[-1]`` for a tuple state;
    returns the tuple at ``t[-1]``.

    Follows torchdiffeq's RKAdaptiveStepsizeODESolver: float64 time, stages evaluated with the time
    cast to the state dtype, one step size for the whole batch chosen from the mixed RMS norm of
    err / (atol + rtol * max(|y0|, |y1|)), safety 0.9, growth <= 10, shrink >= 0.2, initial step by
    Hairer's rule, and the value at ``t[-1]`` read off the 4th-order dense output of the last step.
    Options besides the step-size limits (`_before_integrate` / `_adaptive_step`): ``step_t`` -- times a step must end on;
    ``jump_t`` -- the same, and the derivative is re-evaluated just behind them (``perturb=Perturb.NEXT``: the next
    representable time of the state's dtype); ``norm`` -- a callable on the tuple state replacing the mixed norm.  For a
    decreasing span `_check_inputs` negates the two time lists together with ``t``.
    """
    import bisect
    opts = dict(options or {})
    norm = opts.get("norm") or _tuple_norm
    min_step = float(opts.get("min_step", 0.0))
    max_step = float(opts.get("max_step", float("inf")))
    if method == "dopri5":
        order, ALPHA, BETA, C_SOL, C_ERR, C_MID = 5, _DP5_ALPHA, _DP5_BETA, _DP5_C_SOL, _DP5_C_ERR, _DP5_C_MID
    else:
        order, ALPHA, BETA, C_SOL, C_ERR, C_MID = _ADAPTIVE_TABLEAUX[method]
    t = t.double()
    flip = 1.0
    if bool(t[0] > t[-1]):
        t = -t
        flip = -1.0
        base = func
        func = lambda tt, yy: tuple(-f for f in base(-tt, yy))
    dty = y0[0].dtype
    comb = lambda ks, coefs, scale: tuple(
        sum(k[j] * (coefs[i] * scale) for i, k in enumerate(ks)) for j in range(len(y0)))
    t0 = t[0]
    f0 = func(t0.to(dty), y0)
    # initial step (order - 1 = 4 is what the solver passes); options["first_step"] replaces the rule (torchdiffeq
    # RKAdaptiveStepsizeODESolver._before_integrate)
    first_step = opts.get("first_step")
    max_num_steps = int(opts.get("max_num_steps", 2 ** 31 - 1))
    scale = tuple(atol + a.abs() * rtol for a in y0)
    d0 = norm(tuple(a / s for a, s in zip(y0, scale)))
    d1 = norm(tuple(a / s for a, s in zip(f0, scale)))
    h0 = torch.tensor(1e-6, dtype=dty) if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    h0 = h0.abs()
    y1 = tuple(a + h0 * b for a, b in zip(y0, f0))
    f1 = func(t0.to(dty) + h0, y1)
    d2 = (norm(tuple((a - b) / s for a, b, s in zip(f1, f0, scale))) / h0).abs()
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = torch.max(torch.tensor(1e-6, dtype=dty), h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1, d2)) ** (1.0 / order)
    dt = torch.min(100 * h0, h1.abs()).double()
    if first_step is not None:
        dt = torch.as_tensor(float(first_step), dtype=torch.float64)

    y, f = tuple(y0), f0
    t_lo = t_hi = t0
    last = None
    last_adaptive_stats.update(attempts=0, accepted=0, steps=[])      # steps: (t, dt, error ratio, accepted) per attempt
    n_steps = 0

    def sort_tvals(v):                                            # `_sort_tvals`: the times >= t0, ascending
        if v is None:
            return []
        return sorted(u for u in (flip * torch.as_tensor(v, dtype=torch.float64).reshape(-1)).tolist() if u >= float(t0))
    step_t, jump_t = sort_tvals(opts.get("step_t")), sort_tvals(opts.get("jump_t"))
    if len(set(step_t + jump_t)) != len(step_t + jump_t):
        raise ValueError("`step_t` and `jump_t` must not have any repeated elements between them.")
    i_step = min(bisect.bisect(step_t, float(t0)), len(step_t) - 1)
    i_jump = min(bisect.bisect(jump_t, float(t0)), len(jump_t) - 1)
    while t[-1] > t_hi:
        assert n_steps < max_num_steps, f"max_num_steps exceeded ({n_steps}>={max_num_steps})"
        n_steps += 1
        dt = dt.clamp(min_step, max_step)                         # every attempt starts from a clamped step
        ta, tb = t_hi, t_hi + dt
        if not bool(ta + dt > ta):
            raise AssertionError(f"underflow in dt {float(dt)}")  # torchdiffeq's assertion
        on_step_t = on_jump_t = False
        if step_t:
            on_step_t = bool(ta < step_t[i_step] < ta + dt)
            if on_step_t:
                tb = torch.as_tensor(step_t[i_step], dtype=torch.float64)
                dt = tb - ta
        if jump_t:
            on_jump_t = bool(ta < jump_t[i_jump] < ta + dt)
            if on_jump_t:
                on_step_t = False
                tb = torch.as_tensor(jump_t[i_jump], dtype=torch.float64)
                dt = tb - ta
        ta32, dt32, tb32 = ta.to(dty), dt.to(dty), tb.to(dty)
        ks = [f]
        for alpha, beta in zip(ALPHA, BETA):
            ti = tb32 if alpha == 1.0 else ta32 + alpha * dt32
            yi = tuple(a + b for a, b in zip(y, comb(ks, beta, dt32)))
            ks.append(func(ti, yi))
        y1 = tuple(a + b for a, b in zip(y, comb(ks, C_SOL, dt32)))
        f1 = ks[-1]
        err = comb(ks, C_ERR, dt32)
        tol = tuple(atol + rtol * torch.max(a.abs(), b.abs()) for a, b in zip(y, y1))
        ratio = torch.as_tensor(norm(tuple(e / s for e, s in zip(err, tol)))).abs()
        last_adaptive_stats["attempts"] += 1
        accept = bool(ratio <= 1)
        if dt > max_step:
            accept = False
        if dt <= min_step:
            accept = True
        last_adaptive_stats["steps"].append((float(ta), float(dt), float(ratio), accept))
        if accept:
            ymid = tuple(a + b for a, b in zip(y, comb(ks, C_MID, dt32)))
            last = (ta, tb, dt32, y, y1, ymid, f, f1)
            last_adaptive_stats["accepted"] += 1
            t_lo, t_hi = ta, tb
            y, f = y1, f1
            if on_step_t and i_step != len(step_t) - 1:
                i_step += 1
            if on_jump_t:
                if i_jump != len(jump_t) - 1:
                    i_jump += 1
                f = func(torch.nextafter(tb32, tb32 + 1), y)      # the side of the discontinuity we are on now
        ratio = ratio.double()
        if bool(torch.isnan(ratio)):
            raise AssertionError("underflow in dt nan")       # what torchdiffeq's next attempt asserts
        if ratio == 0:
            dt = dt * 10.0
        else:
            dfactor = 1.0 if ratio < 1 else 0.2
            dt = dt * min(10.0, max(0.9 / float(ratio) ** (1.0 / order), dfactor))
        dt = dt.clamp(min_step, max_step)
    ta, tb, dt32, ya, yb, ymid, fa, fb = last
    x = ((t[-1] - ta) / (tb - ta)).to(dty)
    out = []
    for y_0, y_1, y_m, f_0, f_1 in zip(ya, yb, ymid, fa, fb):
        a = 2 * dt32 * (f_1 - f_0) - 8 * (y_1 + y_0) + 16 * y_m
        b = dt32 * (5 * f_0 - 3 * f_1) + 18 * y_0 + 14 * y_1 - 32 * y_m
        c = dt32 * (f_1 - 4 * f_0) - 11 * y_0 - 5 * y_1 + 16 * y_m
        d = dt32 * f_0
        total = y_0 + x * d
        xp = x
        for coef in (c, b, a):
            xp = xp * x
            total = total + xp * coef
        out.append(total)
    return tuple(out)


def odeint(func, y0, t, method="rk4", options=None, rtol=1e-7, atol=1e-9):
    """Dispatch like torchdiffeq.odeint for the methods restated here."""
    if method == "dopri5" or method in _ADAPTIVE_TABLEAUX:
        return odeint_dopri5(func, y0, t, rtol, atol, options, method)
    return odeint_fixed(func, y0, t, method, options)


# =================================================================================================
# flows  (flow.py)
# =================================================================================================
@dataclass
class FlowParams:
    """state_dict tensors of ODEFlow / ConditionalODEFlow (layers.{0,2,..}.weight/bias, target_*,
    conditional_*)."""
    weights: List[torch.Tensor]
    biases: List[torch.Tensor]
    target_shift: torch.Tensor
    target_scale: torch.Tensor
    conditional_shift: Optional[torch.Tensor] = None
    conditional_scale: Optional[torch.Tensor] = None

    def to(self, dtype):
        f = lambda v: None if v is None else v.to(dtype)
        return FlowParams([w.to(dtype) for w in self.weights], [b.to(dtype) for b in self.biases],
                          f(self.target_shift), f(self.target_scale), f(self.conditional_shift),
                          f(self.conditional_scale))


def flow_params_from_state_dict(sd) -> FlowParams:
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("layers.") and k.endswith(".weight")})
    return FlowParams(weights=[sd[f"layers.{i}.weight"].detach().clone() for i in idx],
                      biases=[sd[f"layers.{i}.bias"].detach().clone() for i in idx],
                      target_shift=sd["target_shift"].detach().clone(), target_scale=sd["target_scale"].detach().clone(),
                      conditional_shift=sd["conditional_shift"].detach().clone() if "conditional_shift" in sd else None,
                      conditional_scale=sd["conditional_scale"].detach().clone() if "conditional_scale" in sd else None)


class FlowOracle:
    """flow.py:9-438 (unconditional) and :441-941 (conditional) restated over plain tensors."""

    def __init__(self, params: FlowParams, dtype=torch.float32, activation=silu):
        self.p = params.to(dtype)
        self.dtype = dtype
        self.activation = activation                                  # one activation() per hidden layer, flow.py:70
        self.twopi = torch.tensor(2.0 * 3.14159265358979323846, dtype=torch.float32).to(dtype)   # flow.py:77

    def velocity(self, inputs):                                       # nn.Sequential, flow.py:68-74
        h = inputs
        for w, b in zip(self.p.weights[:-1], self.p.biases[:-1]):
            h = self.activation(torch.nn.functional.linear(h, w, b))
        return torch.nn.functional.linear(h, self.p.weights[-1], self.p.biases[-1])

    def dynamics(self, t, x, conditional=None):
        """flow.py:112-118 / :580-589: inputs = [x, t, (cond - shift)/scale]."""
        cols = [x, t.view(-1, 1).expand(x.shape[0], 1)]
        if conditional is not None:
            cols.append((conditional - self.p.conditional_shift) / self.p.conditional_scale)
        return self.velocity(torch.cat(cols, dim=1))

    def dynamics_with_jacobian(self, t, x, conditional=None):
        """flow.py:149-161 / :630-644: exact divergence, one autograd call per dimension."""
        with torch.enable_grad():
            x = x.detach().requires_grad_(True)
            dxdt = self.dynamics(t, x, conditional)
            div = torch.zeros(x.shape[0], 1, dtype=x.dtype)
            for i in range(x.shape[-1]):
                div = div + torch.autograd.grad(dxdt[:, i].sum(), x, retain_graph=True)[0][:, i].unsqueeze(1)
        return dxdt.detach(), div.detach()

    def sample(self, xT, conditional=None, method="rk4", options=None, atol=1e-9, rtol=1e-7):
        """flow.py:282-305 / :775-798: integrate t: 1 -> 0, then ``* target_scale + target_shift``
        (the reference passes no tolerances here, so torchdiffeq's defaults apply)."""
        times = torch.tensor([1.0, 0.0], dtype=torch.float32).to(self.dtype)
        with torch.no_grad():
            if conditional is None:
                func = lambda t, y: (self.dynamics(t, y[0]),)
                (x0,) = odeint(func, (xT,), times, method, options, rtol, atol)
            else:      # the conditional rides in the state with a zero derivative, flow.py:591-596, 792-796
                func = lambda t, y: (self.dynamics(t, y[0], y[1]), torch.zeros_like(y[1]))
                x0, _ = odeint(func, (xT, conditional), times, method, options, rtol, atol)
        return x0 * self.p.target_scale + self.p.target_shift

    def solve_ode_forward(self, x, conditional=None, method="rk4", options=None, atol=1e-5, rtol=1e-5):
        """flow.py:347-384 / :844-883 -> (xT, log_jacobian[B,1])."""
        logj = torch.zeros(x.shape[0], 1, dtype=x.dtype)
        times = torch.tensor([0.0, 1.0], dtype=torch.float32).to(self.dtype)
        if conditional is None:
            func = lambda t, y: self.dynamics_with_jacobian(t, y[0])
            return odeint(func, (x, logj), times, method, options, rtol, atol)
        # state (x, conditional, logJ) with d(conditional)/dt = 0, flow.py:646-652, 869-881
        def func(t, y):
            v, div = self.dynamics_with_jacobian(t, y[0], y[1])
            return v, torch.zeros_like(y[1]), div
        xT, _, lj = odeint(func, (x, conditional, logj), times, method, options, rtol, atol)
        return xT, lj

    def log_prob(self, x, conditional=None, method="rk4", options=None, atol=1e-5, rtol=1e-5):
        """flow.py:420-438 / :922-941 -> [B]."""
        x = (x - self.p.target_shift) / self.p.target_scale
        xT, logj = self.solve_ode_forward(x, conditional, method, options, atol, rtol)
        lp = torch.sum(-0.5 * xT ** 2 - 0.5 * torch.log(self.twopi), dim=1)
        return lp + logj.squeeze(1) - torch.sum(torch.log(self.p.target_scale))
