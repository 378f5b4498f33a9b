"""Score-based diffusion models on MI355X: the reference's ``flowfusion.diffusion`` API with the
sampling / log-density solves running as one fused HIP launch.

Class names, constructor arguments, method signatures, return shapes and ``state_dict`` keys
follow ``flowfusion/diffusion.py`` (MLP :9-121, ScoreModel :124-815, VESDE :818-1003,
VPSDE :1006-1180, SUBVPSDE :1183-1366, PopulationModelDiffusion[Conditional] :1466-1848) so a
model trained with the reference loads with ``load_state_dict`` and samples here.

What is native (csrc/ff_mlp_ode.hpp through the C ABI in include/flowfusion_amd.h):
``ScoreModel.sample_ode_from_base``, ``solve_odes_forward`` / ``log_prob`` (Hutchinson probe or
exact trace) and ``sample_sde``, for an ``MLP`` score network with SiLU activations and a
fixed-grid ``method`` (``euler``, ``midpoint``, ``heun3``, ``rk4`` + ``options={"step_size": h}``) or
the reference's default adaptive ``dopri5`` (step control on the device, device_adaptive.py; host controller: adaptive.py),
the divergence by Hutchinson probe, exact trace or the Hutch++ / XTrace estimators (csrc/ff_trace.hip).
CPU tensors raise: there is no eager/CPU fallback behind these methods.  The small pointwise members (``MLP.forward``, ``score``,
``ode_drift``, the SDE schedule functions) are ordinary torch code, used by training code and to
build the per-evaluation tables on the host.  Training losses and the adjoint branches of the
reference are out of scope.
"""
from __future__ import annotations

import copy
import math
import os
from typing import Optional

import torch
from torch import nn
from torch.distributions import Normal

from . import adaptive, device_adaptive, generic, solvers
from . import _native
from . import host_stepper, trace_estimators
from .fused import FusedNet, MODE_EXACT, MODE_HUTCH, MODE_STATE, activation_spec, require_fp32, within_envelope


# ------------------------------------------------------------------------------------------------
# score network
# ------------------------------------------------------------------------------------------------
class MLP(nn.Module):
    """Score network: Gaussian-Fourier time features, then Linear/activation layers.

    Same parameters and buffers as the reference (diffusion.py:32-80): ``NN`` (ModuleList of
    Linear), ``W`` (frozen embedding frequencies, ``embedding_dimensions // 2`` of them drawn as
    ``randn * sigma_initialization``) and ``pi``.  Input column order of the first layer is
    ``[sin, cos, x, conditional]`` (diffusion.py:109-113).
    """

    def __init__(self, n_dimensions=2, n_conditionals=1, embedding_dimensions=8, units=[128],
                 activation=nn.SiLU(), sigma_initialization=16):
        super().__init__()
        self.n_dimensions = n_dimensions
        self.n_conditionals = n_conditionals
        self.architecture = [n_dimensions + n_conditionals + embedding_dimensions] + list(units) + [n_dimensions]
        self.n_layers = len(self.architecture) - 1
        self.NN = nn.ModuleList(
            nn.Linear(n_in, n_out) for n_in, n_out in zip(self.architecture[:-1], self.architecture[1:]))
        self.W = nn.Parameter(torch.randn(embedding_dimensions // 2) * sigma_initialization, requires_grad=False)
        self.activation = activation
        self.register_buffer("pi", torch.tensor(math.pi, dtype=torch.float32))

    def time_features(self, t):
        """[sin(2 pi W t), cos(2 pi W t)] with the reference's fp32 operation order (diffusion.py:109-110)."""
        arg = t[:, None] * self.W[None, :] * 2 * self.pi
        return torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)

    def forward(self, t, x, conditional=None):
        if conditional is not None:
            x = torch.cat([x, conditional], dim=1)
        if t.dim() == 0:
            t = t * torch.ones(x.shape[:-1], device=x.device)
        h = torch.cat([self.time_features(t), x], dim=1)
        for layer in self.NN[:-1]:
            h = self.activation(layer(h))
        return self.NN[-1](h)


# ------------------------------------------------------------------------------------------------
# SDEs (closed-form schedules; reference: diffusion.py:818-1366)
# ------------------------------------------------------------------------------------------------
def _col(v, x):
    return v.view(-1, *[1] * (x.dim() - 1))


class VESDE(nn.Module):
    """Variance-exploding SDE: sigma(t) = sigma_min (sigma_max/sigma_min)^(t/T), zero drift."""

    def __init__(self, sigma_min=1e-2, sigma_max=10.0, T=1.0, epsilon=1e-5):
        super().__init__()
        for name, val in (("T", T), ("epsilon", epsilon), ("sigma_min", sigma_min), ("sigma_max", sigma_max)):
            self.register_buffer(name, torch.tensor(val, dtype=torch.float32))

    def sigma(self, t):
        return self.sigma_min * (self.sigma_max / self.sigma_min) ** (t / self.T)

    def diffusion(self, t, x):
        return _col(self.sigma(t), x) * torch.sqrt(2 * (torch.log(self.sigma_max) - torch.log(self.sigma_min)) / self.T)

    def drift(self, t, x):
        return torch.zeros_like(x)

    def marginal_prob_scalars(self, t):
        return torch.ones_like(t), self.sigma(t)

    def marginal_prob(self, t, x):
        m, s = self.marginal_prob_scalars(t)
        return _col(m, x) * x, _col(s, x)

    def sample_marginal(self, t, x0):
        m, s = self.marginal_prob_scalars(t)
        return _col(m, x0) * x0 + _col(s, x0) * torch.randn_like(x0)

    def prior(self, shape, mu=None):
        if mu is None:
            mu = torch.zeros(shape, device=self.T.device)
        else:
            assert mu.shape == shape
        return Normal(loc=mu, scale=self.sigma_max)


class VPSDE(nn.Module):
    """Variance-preserving SDE with linear beta(t); ``beta_min/beta_max/T`` are plain floats and
    ``epsilon`` a buffer, as in the reference (diffusion.py:1042-1045)."""

    def __init__(self, beta_min=0.1, beta_max=20, T=1.0, epsilon=1e-3):
        super().__init__()
        self.beta_min = beta_min
        self.beta_max = beta_max
        self.T = T
        self.register_buffer("epsilon", torch.tensor(epsilon, dtype=torch.float32))

    def beta(self, t):
        return self.beta_min + (self.beta_max - self.beta_min) * (t / self.T)

    def _log_coeff(self, t):
        return 0.5 * (self.beta_max - self.beta_min) * t ** 2 / self.T + self.beta_min * t

    def marginal_prob_scalars(self, t):
        lc = self._log_coeff(t)
        return torch.exp(-0.5 * lc), torch.sqrt(1.0 - torch.exp(-lc))

    def sigma(self, t):
        return self.marginal_prob_scalars(t)[1]

    def prior(self, shape):
        return Normal(loc=torch.zeros(shape, device=self.epsilon.device), scale=1.0)

    def diffusion(self, t, x):
        return _col(torch.sqrt(self.beta(t)), x)

    def drift(self, t, x):
        return -0.5 * _col(self.beta(t), x) * x

    def marginal_prob(self, t, x):
        m, s = self.marginal_prob_scalars(t)
        return _col(m, x) * x, _col(s, x)


class SUBVPSDE(VPSDE):
    """Sub-VP SDE: same drift as VP, g^2 = beta (1 - exp(-2 int beta)), std = 1 - exp(-int beta)."""

    def diffusion(self, t, x):
        decay = torch.exp(-2 * self.beta_min * t - (self.beta_max - self.beta_min) * t ** 2 / self.T)
        return _col(torch.sqrt(self.beta(t) * (1.0 - decay)), x)

    def marginal_prob_scalars(self, t):
        lc = self._log_coeff(t)
        return torch.exp(-0.5 * lc), 1.0 - torch.exp(-lc)


# ------------------------------------------------------------------------------------------------
# score model
# ------------------------------------------------------------------------------------------------
class ScoreModel(nn.Module):
    """Sampler / density evaluator for a score network and an SDE (reference: diffusion.py:124-815)."""

    def __init__(self, model=None, sde=None, conditional=None, no_sigma=False, hutchinson=False,
                 hutchpp=False, hpp_rank=1, hpp_vecs=1, xtrace=False, xt_vecs=1, *, precision="f32"):
        """Arguments as in the reference (diffusion.py:158-170).  Extension, keyword only: ``precision`` (also an
        attribute that can be flipped later) selects the arithmetic of the Linear layers in the fused solves:
        ``"f32"`` (default) is exact fp32, what the reference computes; ``"bf16x3"`` runs them on the bf16 matrix cores
        with every operand split into three bf16 parts and six products per term -- fp32-class accuracy (1e-7
        relative per layer) at about twice the speed, for the state-only solves (sampling on fixed grids and with the
        adaptive methods, ``sample_sde``) of SiLU networks with 1-4 hidden layers up to 256 wide, dim <= 16.
        ``"bf16x2"``: two bf16 parts by round-to-nearest (operands to 16 significand bits, unbiased) and three products per
        term -- half the matrix work of ``"bf16x3"``; a layer's error is ~4e-7 relative in the mean (fp32: 2e-8), end to
        end the sampler and the log-density of the headline configuration stay at the fp32 kernel's distance from the
        float64 oracle; same networks, also the Hutchinson and exact-trace log-densities and, state-only, dim <= 32.
        Anything else -- the Hutch++ / XTrace estimators, other activations, wider or deeper networks -- raises with these
        settings (round 3 froze the family at what the BASELINE configurations and the reference's demos reach)."""
        super().__init__()
        self.precision = precision
        self.model = model
        self.sde = sde
        self.conditional = conditional
        self.no_sigma = no_sigma
        self.prob = False
        self.hutch = hutchinson
        self.hutchpp = hutchpp
        self.hpp_rank = hpp_rank
        self.hpp_vector = hpp_vecs
        self.xtrace = xtrace
        self.xt_vector = xt_vecs
        self._fused = None

    # -- pointwise pieces (plain torch) -------------------------------------------------------
    def score(self, t, x, conditional=None):
        out = self.model(t, x, conditional=conditional)
        if self.no_sigma:
            return out
        return out / _col(self.sde.sigma(t), x)

    def ode_drift(self, t, x, conditional=None):
        g = self.sde.diffusion(t, x)
        return self.sde.drift(t, x) - 0.5 * g ** 2 * self.score(t, x, conditional=conditional)

    def forward(self, t, states):
        """ODE right-hand side in torchdiffeq's calling convention: ``states = (x,)`` or
        ``(x, dlogp)``; returns the matching tuple of time derivatives (reference :281-508).
        Autograd-based and differentiable; the fused solves do not call it."""
        x = states[0]
        if not self.prob:
            with torch.set_grad_enabled(True):
                x.requires_grad_(True)
                return self.ode_drift(t, x, conditional=self.conditional)
        n = x.shape[0]
        with torch.set_grad_enabled(True):
            x.requires_grad_(True)
            xdot = self.ode_drift(t, x, conditional=self.conditional)
            if self.hutch:
                vjp = torch.autograd.grad(xdot, x, self.e, create_graph=True, retain_graph=True)[0]
                div = (vjp * self.e).sum(dim=1)
            elif self.hutchpp or self.xtrace:
                # the reference detaches the estimator's products (diffusion.py:375-396): not differentiable
                rows = [torch.autograd.grad(xdot[:, i].sum(), x, retain_graph=True)[0] for i in range(x.shape[1])]
                A = torch.stack(rows, dim=1).transpose(1, 2).contiguous()          # A[b] = J[b]^T
                div = self._estimate_divergence(A.detach(), x)
            else:
                div = x.new_zeros(n)
                for i in range(x.shape[1]):
                    div = div + torch.autograd.grad(xdot[:, i].sum(), x, create_graph=True, retain_graph=True)[0][:, i]
        return xdot, div.view(n, 1)

    # -- Hutch++ / XTrace (reference :336-481) ---------------------------------------------------
    def _probe_counts(self, D):
        """(r, m) of Hutch++ and m of XTrace with the reference's clamps (:343-344, :409)."""
        return (int(min(self.hpp_rank, D)), int(max(1, self.hpp_vector))), int(min(max(1, self.xt_vector), D))

    def _estimate_divergence(self, A, x):
        """Divergence estimate [B] from A[b] = J[b]^T with the stored probes (drawn afresh, like the reference's
        fallback :350-353, :415-416, when none of the right shape are stored).  On the GPU the estimate is one launch of
        ff_trace_estimate (csrc/ff_trace.hip); CPU tensors -- ``forward`` in training code, the tests' emulator -- and
        ``FF_TORCH_ESTIMATOR=1`` (A/B runs) take the torch statement of the same formulas (trace_estimators.py)."""
        B, D = x.shape
        (r, m), mx = self._probe_counts(D)
        native = A.is_cuda and A.dtype == torch.float32 and os.environ.get("FF_TORCH_ESTIMATOR", "") in ("", "0")
        if self.hutchpp:
            S, G = getattr(self, "S", None), getattr(self, "G", None)
            if S is None or tuple(S.shape) != (r, B, D):
                S = trace_estimators.draw_probes(r, x)
            if G is None or tuple(G.shape) != (m, B, D):
                G = trace_estimators.draw_probes(m, x)
            if native:
                return _native.trace_estimate(A.unsqueeze(0), "hutchpp", (S, G))[0]
            return trace_estimators.hutchpp(A, S.to(A.device), G.to(A.device))
        O = getattr(self, "O", None)
        if O is None or tuple(O.shape) != (mx, B, D):
            O = trace_estimators.draw_probes(mx, x)
        if native:
            return _native.trace_estimate(A.unsqueeze(0), "xtrace", (O,))[0]
        return trace_estimators.xtrace(A, O.to(A.device))

    # -- any other `model=` module: native stepping around the module's own forward (generic.py) -------------
    def _fusable(self) -> bool:
        """The score network is the reference's MLP and a compiled kernel holds its shape and activation.  Anything
        else -- a user module, or an MLP wider / higher-dimensional than the compiled shapes or with an activation
        the kernels do not implement -- is stepped by generic.py (the reference has no such limit, diffusion.py:59-72).
        An explicit ``precision=`` other than f32 never switches arithmetic silently: ``_net()`` raises instead."""
        m = self.model
        if not (hasattr(m, "NN") and hasattr(m, "W") and hasattr(m, "pi") and hasattr(m, "n_dimensions")):
            return False
        if getattr(self, "precision", "f32") != "f32":
            return True
        key = tuple(id(l) for l in m.NN) + (repr(m.activation),)
        return within_envelope(self, key, self._net)

    def _rhs_module(self, t, y):
        """ODE right-hand side through ``self.forward`` (the reference's own RHS, diffusion.py:281-508): the user's
        module evaluates the score; divergences come from autograd exactly as in the reference."""
        if not self.prob:
            return self.forward(t, (y,)), None
        xdot, div = self.forward(t, (y, None))
        return xdot, div.reshape(-1)

    def _solve_generic(self, x, t_span, method, options, mode, atol, rtol, affine):
        if affine.get("in_shift") is not None:
            x = (x - affine["in_shift"]) / affine["in_scale"]
        y, lp, stats = generic.solve(self._rhs_module, x, t_span, method, options, mode != MODE_STATE, atol, rtol)
        self.last_solver_stats = stats
        if affine.get("out_scale") is not None:
            y = y * affine["out_scale"] + affine["out_shift"]
        return y, lp

    # -- fused path -----------------------------------------------------------------------------
    def _net(self) -> FusedNet:
        m = self.model
        if not (hasattr(m, "NN") and hasattr(m, "W") and hasattr(m, "pi") and hasattr(m, "n_dimensions")):
            raise NotImplementedError(
                "the fused gfx950 path needs a flowfusion MLP score network "
                f"(NN/W/pi attributes); got {type(m).__name__}")
        act = activation_spec(m.activation)
        prec = getattr(self, "precision", "f32")
        if self._fused is None or not self._fused.serves(list(m.NN), act, prec):
            E = 2 * m.W.numel()
            self._fused = FusedNet(list(m.NN), m.n_dimensions, m.n_conditionals, x_col0=E,
                                   c_col0=E + m.n_dimensions, act=act, precision=prec)
        return self._fused

    def _device_schedule(self, device):
        """What the device-side adaptive controller needs to evaluate this model's time-dependent terms itself
        (``device_adaptive.ScheduleSpec``), or None when it cannot: an SDE class other than the reference's three, or a
        score network without the MLP's Fourier features.  Scalars that live in buffers are read once per value."""
        sde, m = self.sde, self.model
        kind = {VESDE: _native.SCHED_VE, VPSDE: _native.SCHED_VP, SUBVPSDE: _native.SCHED_SUBVP}.get(type(sde))
        if kind is None or not (hasattr(m, "W") and hasattr(m, "pi") and hasattr(m, "NN")):
            return None
        names = ("sigma_min", "sigma_max", "T") if kind == _native.SCHED_VE else ("beta_min", "beta_max", "T")
        vals = [getattr(sde, n) for n in names] + [m.pi]
        key = tuple((v.data_ptr(), v._version) if torch.is_tensor(v) else v for v in vals)
        hit = self.__dict__.get("_dev_sched_scalars")
        if hit is None or hit[0] != key:
            hit = (key, [float(v) for v in vals])
            object.__setattr__(self, "_dev_sched_scalars", hit)
        p0, p1, p2, pi = hit[1]
        E = 2 * m.W.numel()
        w0t, b0 = self._net().time_columns(device, 0, E)
        emb_w = m.W.detach().to(device, torch.float32).contiguous()
        return device_adaptive.ScheduleSpec(kind, (p0, p1, p2), bool(self.no_sigma), emb_w, pi, w0t, b0)

    def _schedule_inputs(self):
        """Host copies of everything `_schedule` reads (SDE, embedding frequencies, first layer).  Taken once
        per solve: the adaptive driver calls `_schedule` at every attempted step, and each device-to-host copy
        would wait for the kernel in flight."""
        m = self.model
        w0, b0 = self._net().first_layer_cpu()
        return (copy.deepcopy(self.sde).to("cpu"), m.W.detach().to("cpu", torch.float32),
                m.pi.detach().to("cpu", torch.float32), w0, b0)

    def _schedule(self, t: torch.Tensor, sde_form: str, host=None):
        """Per-evaluation scalars (a, b) and first-layer bias c1 for real times ``t`` (fp32, CPU).

        ODE (diffusion.py:276-278): xdot = f - 0.5 g^2 score  ->  a = f/x, b = -0.5 g^2 [/ sigma]
        reverse SDE (:553):         f - g^2 score             ->  b = -g^2 [/ sigma]
        with f = a(t) x for all three SDEs (:905, :1131, :1316).  Returns also g (for the noise).
        """
        sde, W, pi, w0, b0 = host if host is not None else self._schedule_inputs()
        with solvers.host_threads():
            return self._schedule_on_host(t, sde_form, sde, W, pi, w0, b0)

    def _schedule_on_host(self, t, sde_form, sde, W, pi, w0, b0):
        one = torch.ones(t.numel(), 1, dtype=torch.float32)
        a = sde.drift(t, one).reshape(-1)
        g = sde.diffusion(t, one).reshape(-1)
        b = -(0.5 * g ** 2) if sde_form == "ode" else -(g ** 2)
        if not self.no_sigma:
            b = b / sde.sigma(t).reshape(-1)
        arg = t[:, None] * W[None, :] * 2 * pi
        emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
        # broadcast product + sum rather than a matmul: a BLAS call this small costs ~0.5-8 ms when it wakes a
        # 128-thread pool on a many-core host (measured), the elementwise form stays on the calling thread
        c1 = (emb[:, None, :] * w0[None, :, : emb.shape[1]]).sum(-1) + b0
        return a, b, c1, g

    def _check_inputs(self, x, what):
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError(
                f"{what}: gradients through the fused solve are not available (the reference's "
                "odeint_adjoint branch, diffusion.py:620-629, is out of scope); detach the input")

    def _ode_table(self, t_span, method, options, mode, y0=None):
        plan = solvers.plan_ode(t_span, method, options, y0=y0)
        self._net().require_slots(int(plan.slot.max()) + 1, mode, f"method={method!r}")
        a, b, c1, _ = self._schedule(plan.t_eval, "ode")
        return solvers.build_table(plan, a, b, c1, self._net().width(mode))

    def _solve(self, x, t_span, method, options, mode, atol, rtol, cond=None, probe=None, **affine):
        """odeint(self, state, t_span, method=, atol=, rtol=, options=) on the fused kernels:
        fixed-grid methods as one launch, the adaptive methods with the step control on the device.  ``affine``
        (in_shift / in_scale / out_scale / out_shift, the PopulationModel wrappers' pre- and post-processing)
        rides in the kernel's prologue / epilogue on fixed grids and is applied around the adaptive loop."""
        require_fp32(self, x, cond, probe, what="an ODE solve")
        if not self._fusable():
            return self._solve_generic(x, t_span, method, options, mode, atol, rtol, affine)
        net = self._net()
        if method in solvers.ALL_ADAPTIVE:
            if affine.get("in_shift") is not None:
                x = (x - affine["in_shift"]) / affine["in_scale"]
            t = t_span.detach().to("cpu", torch.float32).double()
            sign = -1.0 if bool(t[0] > t[-1]) else 1.0
            spec = self._device_schedule(x.device) if (x.is_cuda and method in solvers.NATIVE_ADAPTIVE) else None
            if device_adaptive.supported(spec, x, net, mode, options):
                # the whole loop on the device: attempts, error norms, step control, the next attempt's table rows
                y, lp, stats = device_adaptive.solve(net, spec, sign, mode, x, float(sign * t[0]), float(sign * t[-1]),
                                                     rtol, atol, options, method, cond=cond, probe=probe)
                self.last_solver_stats = stats
                if affine.get("out_scale") is not None:
                    y = y * affine["out_scale"] + affine["out_shift"]
                return y, lp
            host = self._schedule_inputs()
            sched = lambda tr: self._schedule(tr, "ode", host)[:3]
            step = net.make_step(sched, sign, mode, x.device, cond=cond, probe=probe)
            solver = adaptive.make_solver(step, mode != MODE_STATE, rtol, atol, options, method=method, sign=sign)
            lp0 = torch.zeros(x.shape[0], device=x.device) if mode != MODE_STATE else None
            y, lp = solver.integrate(float(sign * t[0]), float(sign * t[-1]),
                                     x.detach().to(torch.float32).contiguous(), lp0)
            self.last_solver_stats = {"attempts": solver.n_attempts, "accepted": solver.n_accepted}
            if affine.get("out_scale") is not None:
                y = y * affine["out_scale"] + affine["out_shift"]
            return y, lp
        key = ("score-ode", tuple(float(v) for v in t_span), method, repr(sorted((options or {}).items())), mode,
               self.no_sigma, self._schedule_key())
        if (options or {}).get("grid_constructor") is not None:      # the grid may depend on y0: built per call
            table = self._ode_table(t_span, method, options, mode, y0=x).to(x.device)
        else:
            table = net.cached_table(key, x.device, lambda: self._ode_table(t_span, method, options, mode))
        y, lp, _ = net.integrate(x, table, mode, cond=cond, probe=probe, stage_slots=solvers.resolve_method(method).stages, **affine)
        return y, (lp if mode != MODE_STATE else None)

    def _schedule_key(self):
        """Everything besides the first layer that the evaluation table depends on."""
        sde = self.sde
        vals = [type(sde).__name__]
        for name in ("beta_min", "beta_max", "T", "epsilon", "sigma_min", "sigma_max"):
            if hasattr(sde, name):
                vals.append(float(getattr(sde, name)))
        m = self.model
        return tuple(vals) + (m.W.data_ptr(), m.W._version)

    @torch.no_grad()
    def sample_sde(self, shape, conditional=None, steps=100, *, noise="torch", seed=None, sample_offset=0,
                   progress=None, progress_every=None):
        """Euler-Maruyama sampling of the reverse SDE; returns the last *mean* state, like the
        reference (diffusion.py:510-563).  By default random numbers are drawn exactly as the reference
        draws them on the model's device (one prior draw, then one ``randn_like`` per step), so a given
        ``torch.manual_seed`` reproduces the reference's stream on that device.  If a step produces a NaN the
        reference prints a message, stops and returns that step's mean (:560-563); so does this method (the
        step is located after the fact, by re-running a prefix of the launch that reported it).  One difference after
        such a stop: the reference stops DRAWING at the failing step, here the normals of the whole chunk (and, with two
        chunks in flight, of the next one) have been drawn by then, so torch's generator is further along than the
        reference's would be -- the returned tensor is the reference's, the generator state afterwards is not.

        Extensions: ``noise="philox"`` draws the per-step normals inside the kernel (counter-based, keyed by
        ``seed`` and the global sample index ``sample_offset + row``; include/flowfusion_amd.h): no noise
        buffers, no random-number kernels, and the result for a sample does not depend on how the batch is
        split over launches or GPUs (``flowfusion_amd.distributed.sample_sde_sharded``).  ``seed=None``
        takes one from torch's default generator, so ``torch.manual_seed`` still fixes the run.
        ``progress(done, total)`` is called after every ``progress_every`` steps (default: a tenth of the
        run) -- the reference's tqdm bar (:543-547) at launch granularity instead of three host syncs per
        step."""
        batch, *dims = shape
        dev = next(self.model.parameters()).device
        x = self.sde.prior(dims).sample([batch]).to(dev)
        kw = dict(progress=progress, progress_every=progress_every)
        if noise == "torch":
            return self._sample_sde_from(x, lambda like: torch.randn_like(like), conditional, steps, **kw)
        if noise != "philox":
            raise ValueError(f"noise={noise!r}: expected 'torch' or 'philox'")
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        return self._sample_sde_from(x, None, conditional, steps, rng=(int(seed), int(sample_offset)), **kw)

    @torch.no_grad()
    def _sample_sde_from(self, x, draw, conditional=None, steps=100, rng=None, progress=None, progress_every=None):
        """The Euler-Maruyama loop proper: ``x`` is the prior draw, ``draw(like)`` supplies the i-th
        standard-normal slab (tests inject the reference's captured stream here); with
        ``rng = (seed, global index of row 0)`` the kernel draws the normals itself.  The steps run in as few
        launches as the noise memory and the progress granularity allow (one, normally)."""
        require_fp32(self, x, conditional, what="sample_sde")
        if not self._fusable():
            if rng is not None:
                raise NotImplementedError("noise='philox' lives in the fused kernel; a custom score module samples with noise='torch'")
            host = copy.deepcopy(self.sde).to("cpu")
            drift = lambda t, xx: self.sde.drift(t, xx) - self.sde.diffusion(t, xx) ** 2 * self.score(t, xx, conditional=conditional)
            g_of_t = lambda ts: host.diffusion(ts, torch.ones(ts.numel(), 1))
            return generic.euler_maruyama(drift, g_of_t, x, draw, torch.as_tensor(self.sde.T, dtype=torch.float32).cpu(),
                                          self.sde.epsilon.detach().cpu(), steps, progress)
        net = self._net()
        if x.dim() != 2:
            raise NotImplementedError("sample_sde: only [batch, dim] states are supported")
        dev = x.device
        batch = x.shape[0]
        T = torch.as_tensor(self.sde.T, dtype=torch.float32).cpu()
        eps = self.sde.epsilon.detach().cpu()
        ts, dt = solvers.plan_euler_maruyama(T, eps, steps)
        n = int(ts.numel())
        if n == 0:
            raise RuntimeError("sample_sde: T < epsilon, no step to take")
        a, b, c1, g = self._schedule(ts, "sde")
        gn = g * (-dt) ** (1.0 / 2.0)                 # g * sqrt(-dt)  (:554-558)
        cout = torch.zeros(n, 8)
        cout[:, 0] = dt                               # x_mean = x + f dt  (:557)
        zeros8 = torch.zeros(n, 8)
        slot = torch.zeros(n, dtype=torch.int32)
        width = net.width(MODE_STATE)

        def table(start, stop, mean_last):
            """Rows of steps [start, stop); with `mean_last` the last one returns x_mean (no noise added, :563)."""
            flags = torch.full((stop - start,), solvers.FLAG_STEP_END | solvers.FLAG_NOISE, dtype=torch.int32)
            if mean_last:
                flags[-1] = solvers.FLAG_STEP_END
            sub = solvers.EvalPlan(t_eval=ts[start:stop], sign=1.0, slot=slot[start:stop], flags=flags,
                                   cin=zeros8[start:stop], cout=cout[start:stop], n_steps=stop - start)
            return solvers.build_table(sub, a[start:stop], b[start:stop], c1[start:stop], width,
                                       gn=gn[start:stop], noise_idx=torch.arange(stop - start)).to(dev)

        def launch(x_in, start, stop, mean_last, buf):
            if rng is not None:
                return net.integrate(x_in, table(start, stop, mean_last), MODE_STATE, cond=conditional,
                                     rng=(rng[0] & 0x7FFFFFFFFFFFFFFF, rng[1], start), stage_slots=1)
            return net.integrate(x_in, table(start, stop, mean_last), MODE_STATE, cond=conditional,
                                 noise=buf[: stop - start], stage_slots=1)

        def first_nan_mean(x_in, start, stop, buf):
            """A launch over steps [start, stop) reported NaN: the reference would have stopped at the first
            step whose state x (noise included, :558-560) has one and returned that step's mean.  NaN is
            absorbing, so bisect on the number of full steps (each probe = one launch of a prefix), then
            re-run that prefix with a mean-only last row."""
            lo, hi = 1, stop - start                  # smallest m with a NaN in x after m steps
            while lo < hi:
                mid = (lo + hi) // 2
                if int(launch(x_in, start, start + mid, False, buf)[2].item()) & 1:
                    hi = mid
                else:
                    lo = mid + 1
            return launch(x_in, start, start + lo, True, buf)[0]

        # chunk boundaries: noise slabs are drawn per step, in order, at most 2^28 floats (1 GiB) per buffer;
        # a progress callback adds boundaries of its own
        per_step = max(batch * x.shape[1], 1)
        chunk = n if rng is not None else max(1, min(n, (1 << 28) // per_step))
        if progress is not None:
            every = int(progress_every) if progress_every else max(1, n // 10)
            chunk = max(1, min(chunk, every))
        bounds = [(s0, min(n, s0 + chunk)) for s0 in range(0, n, chunk)]
        bufs = [] if rng is not None else [
            torch.empty(min(chunk, n), batch, x.shape[1], device=dev, dtype=torch.float32)
            for _ in range(min(2, len(bounds)))]
        # The draws of chunk c+1 are enqueued on a side stream while the kernel integrates chunk c (two
        # buffers), so the random-number kernels stay off the critical path; the host-side generator is
        # advanced in the same order either way, so the stream of numbers is unchanged.
        overlap = rng is None and len(bounds) > 1
        main = torch.cuda.current_stream(dev) if x.is_cuda else None
        side = torch.cuda.Stream(device=dev) if overlap else None
        ready = [None] * len(bounds)
        freed = [None] * len(bufs)

        def fill(c):
            start, stop = bounds[c]
            buf = bufs[c % len(bufs)]
            if not overlap:
                for i in range(stop - start):                   # one draw per executed step, the last included (:554)
                    buf[i] = draw(x)
                return
            with torch.cuda.stream(side):
                if freed[c % len(bufs)] is not None:
                    side.wait_event(freed[c % len(bufs)])       # the kernel that read this buffer is done
                else:
                    side.wait_stream(main)
                for i in range(stop - start):
                    buf[i] = draw(x)
                ready[c] = side.record_event()

        if rng is None:
            fill(0)
        for c, (start, stop) in enumerate(bounds):
            if overlap and c + 1 < len(bounds):
                fill(c + 1)
            if overlap:
                main.wait_event(ready[c])
            buf = bufs[c % len(bufs)] if bufs else None
            x_in = x
            x, _, status = launch(x_in, start, stop, stop == n, buf)
            if overlap:
                freed[c % len(bufs)] = main.record_event()
            if int(status.item()) & 1:
                print("Diffusion is not stable, NaN were produced. Stopped sampling.")
                return first_nan_mean(x_in, start, stop, buf)
            if progress is not None:
                progress(stop, n)
        return x

    def sample_ode_from_base(self, base_samples, conditional=None, atol=1e-4, rtol=1e-4,
                             method="dopri5", options=None):
        """Probability-flow ODE from t=1 down to epsilon; returns ``(samples, [])`` like the
        reference (diffusion.py:566-640).  ``atol``/``rtol`` are accepted for signature
        compatibility and unused by the fixed-grid methods."""
        return self._sample_ode(base_samples, conditional, atol, rtol, method, options), []

    def _sample_ode(self, base_samples, conditional, atol, rtol, method, options, out_scale=None, out_shift=None):
        """sample_ode_from_base proper; ``x * out_scale + out_shift`` (PopulationModel*.forward,
        diffusion.py:1575-1585, 1772-1784) is applied by the kernel's epilogue."""
        self._check_inputs(base_samples, "sample_ode_from_base")
        if self._fusable():
            self._net()
        z = base_samples * self.sde.sigma_max if hasattr(self.sde, "sigma_max") else base_samples
        self.prob = False
        self.conditional = conditional
        t_span = torch.tensor([1.0, float(self.sde.epsilon)], dtype=torch.float32)
        x, _ = self._solve(z, t_span, method, options, MODE_STATE, atol, rtol, cond=conditional,
                           out_scale=out_scale, out_shift=out_shift)
        return x

    @torch.no_grad()
    def solve_odes_forward(self, x0_samples, conditional=None, atol=1e-5, rtol=1e-5,
                           method="dopri5", options=None, *, probe="torch", seed=None, sample_offset=0):
        """Probability-flow ODE from epsilon up to t=1 with the divergence integrated alongside;
        returns ``(xT, delta_logp[B,1])`` (reference: diffusion.py:642-754).

        Extension, keyword only (Hutchinson models): ``probe="torch"`` (default) draws the +-1 probe on the CPU and
        moves it, exactly as the reference does (:701), so ``torch.manual_seed`` reproduces its stream;
        ``probe="philox"`` takes the signs of the library's counter-based normals keyed by ``seed`` and the GLOBAL row
        ``sample_offset + r`` (``ff_normal_fill`` with the reserved probe index): drawn on the device (no host draw, no
        upload) and independent of how a batch is cut into shards (``distributed.log_prob_sharded``)."""
        return self._solve_forward(x0_samples, conditional, atol, rtol, method, options,
                                   probe_rng=self._probe_rng(probe, seed, sample_offset))

    def _probe_rng(self, probe, seed, sample_offset):
        if probe == "torch":
            if seed is not None:
                raise ValueError("seed= belongs to probe='philox' (the torch probe follows torch.manual_seed)")
            return None
        if probe != "philox":
            raise ValueError(f"probe must be 'torch' or 'philox', not {probe!r}")
        if not (self.hutch or self.hutchpp or self.xtrace):
            raise ValueError("probe='philox' draws the probes of a Hutchinson / Hutch++ / XTrace model: construct it with "
                             "hutchinson=True, hutchpp=True or xtrace=True")
        if seed is None:      # one draw of torch's generator, so torch.manual_seed still fixes the run
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        return int(seed), int(sample_offset)

    @torch.no_grad()
    def _solve_forward(self, x0_samples, conditional, atol, rtol, method, options, in_shift=None, in_scale=None,
                       probe_rng=None):
        """solve_odes_forward proper; ``(x - in_shift) / in_scale`` (PopulationModel*.log_prob,
        diffusion.py:1633, 1837) is applied by the kernel's prologue."""
        fused = self._fusable()
        if fused:
            self._net()
        self.prob = True
        self.conditional = conditional
        if (self.hutchpp or self.xtrace) and not self.hutch:
            if in_shift is not None:
                x0_samples = (x0_samples - in_shift) / in_scale
            if fused:
                return self._solve_with_estimator(x0_samples, conditional, atol, rtol, method, options, probe_rng)
            if probe_rng is not None:
                raise NotImplementedError("probe='philox' lives on the fused path; a custom score module draws its probes with torch")
            # any other module: forward() runs the estimator itself (reverse mode, like the reference); the probes are
            # drawn once per solve on the state's device (:703-719)
            (r, m), mx = self._probe_counts(x0_samples.shape[1])
            if self.hutchpp:
                self.S = trace_estimators.draw_probes(r, x0_samples)
                self.G = trace_estimators.draw_probes(m, x0_samples)
            else:
                self.O = trace_estimators.draw_probes(mx, x0_samples)
            t_span = torch.tensor([float(self.sde.epsilon), 1.0], dtype=torch.float32)
            xT, dlogp = self._solve_generic(x0_samples, t_span, method, options, MODE_EXACT, atol, rtol, {})
            return xT, dlogp.view(-1, 1)
        probe = None
        mode = MODE_EXACT
        if self.hutch and probe_rng is not None:
            if x0_samples.dim() != 2:
                raise NotImplementedError("probe='philox': only [batch, dim] states")
            z = _native.normal_fill(x0_samples.shape[0], x0_samples.shape[1], probe_rng[0], probe_rng[1],
                                    x0_samples.device, noise_index=_native.PROBE_NOISE_INDEX)
            self.e = torch.where(z >= 0, 1.0, -1.0).to(torch.float32)
        elif self.hutch:
            # drawn on the CPU and moved, as the reference does (diffusion.py:701)
            self.e = torch.sign(torch.randn(x0_samples.shape)).to(x0_samples.device)
        if self.hutch:
            probe = self.e
            mode = MODE_HUTCH
        t_span = torch.tensor([float(self.sde.epsilon), 1.0], dtype=torch.float32)
        xT, dlogp = self._solve(x0_samples, t_span, method, options, mode, atol, rtol, cond=conditional, probe=probe,
                                in_shift=in_shift, in_scale=in_scale)
        return xT, dlogp.view(-1, 1)

    def _solve_with_estimator(self, x0, conditional, atol, rtol, method, options, probe_rng=None):
        """Hutch++ / XTrace log-density solve.  The state never depends on the divergence, so the launches are those of the
        exact trace with the Jacobian of every evaluation row recorded (ff_ode_args.jac_all); the estimates of all rows
        come from ONE launch (ff_trace_estimate, csrc/ff_trace.hip) and are combined with the tableau's weights.  The
        adaptive methods run with the step control on the device like every other solve (device_adaptive.py); the host
        controller with one launch per right-hand side (host_stepper.py) stays for what that cannot describe.  Probes are
        drawn once per solve on the state's device, as the reference does (:703-719)."""
        net = self._net()
        if net.precision != "f32":
            raise NotImplementedError(f"precision={net.precision!r}: the Hutch++ / XTrace estimators need the Jacobian output of the "
                                      "f32 kernels (bf16x2 serves hutchinson=True and the exact trace)")
        if not x0.is_cuda:
            raise RuntimeError("flowfusion_amd integrates on the GPU only: move the model and its inputs to 'cuda' "
                               f"(got a tensor on {x0.device}); there is no CPU fallback")
        B, D = x0.shape
        (r, m), mx = self._probe_counts(D)
        if probe_rng is not None:       # probe="philox": keyed by (seed, global row), independent of the sharding
            draw = lambda n, second=False: trace_estimators.draw_probes_philox(n, x0, probe_rng[0], probe_rng[1], second)
        else:
            draw = lambda n, second=False: trace_estimators.draw_probes(n, x0)
        if self.hutchpp:
            self.S = draw(r)
            self.G = draw(m, True)
            kind, probes = "hutchpp", (self.S, self.G)
        else:
            # the reference stores max(1, xt_vecs) probes and redraws inside forward when that exceeds D (:719, :409-416)
            self.O = draw(mx)
            kind, probes = "xtrace", (self.O,)
        x = x0.detach().to(torch.float32).contiguous()
        t_span = torch.tensor([float(self.sde.epsilon), 1.0], dtype=torch.float32)
        if method in solvers.ALL_ADAPTIVE:
            spec = self._device_schedule(x.device) if method in solvers.NATIVE_ADAPTIVE else None
            if device_adaptive.supported(spec, x, net, MODE_EXACT, options) and \
                    device_adaptive.estimator_bytes(net, method, B, kind, probes) <= self._estimator_budget(x.device):
                y, lp, stats = device_adaptive.solve(net, spec, 1.0, MODE_EXACT, x, float(t_span[0]), float(t_span[1]), rtol, atol,
                                                     options, method, cond=conditional, estimator=(kind, probes))
                self.last_solver_stats = stats
                return y, lp.view(-1, 1)
            stepper = host_stepper.RowStepper(net, x.device, conditional, lambda A: self._estimate_divergence(A, x))
            host = self._schedule_inputs()
            sched = lambda tr: self._schedule(tr, "ode", host)[:3]
            solver = adaptive.make_solver(stepper.make_step(sched, 1.0), True, rtol, atol, options, method=method)
            y, lp = solver.integrate(float(t_span[0]), float(t_span[1]), x, torch.zeros(B, device=x.device))
            self.last_solver_stats = {"attempts": solver.n_attempts, "accepted": solver.n_accepted}
        else:
            # fixed grid: one launch per tangent pass records every row's Jacobian, one launch estimates all of them
            stepper = host_stepper.RowStepper(net, x.device, conditional, None)
            div_rows = lambda A, lo, hi: _native.trace_estimate(A, kind, tuple(P[:, lo:hi] for P in probes))
            # (the recorded Jacobians of a chunk of samples may take a quarter of the free device memory: cutting a batch
            # into many small launches leaves their last rounds of tiles mostly empty)
            y, lp = stepper.run_table_recorded(x, self._ode_table(t_span, method, options, MODE_EXACT), div_rows,
                                               cond=conditional, max_bytes=max(1 << 30, self._estimator_budget(x.device) // 2))
        return y, lp.view(-1, 1)

    @staticmethod
    def _estimator_budget(device) -> int:
        """Bytes the recorded Jacobians of an attempted step may take: half of what the device has free right now."""
        free, _ = torch.cuda.mem_get_info(device)
        return free // 2

    @torch.no_grad()
    def log_prob(self, x0_samples, conditional=None, atol=1e-4, rtol=1e-4, method="dopri5",
                 options={"min_step": 1e-6}, *, probe="torch", seed=None, sample_offset=0):
        """log p(x0) = delta_logp + log prior(xT), shape [B,1] (reference: diffusion.py:756-815).  ``probe`` / ``seed`` /
        ``sample_offset``: see ``solve_odes_forward``."""
        xT, lp = self.solve_odes_forward(x0_samples, conditional=conditional, atol=atol, rtol=rtol,
                                         method=method, options=options, probe=probe, seed=seed,
                                         sample_offset=sample_offset)
        return lp + torch.sum(self.sde.prior(xT.shape).log_prob(xT), dim=1, keepdim=True)


# ------------------------------------------------------------------------------------------------
# population-model wrappers (affine pre/post-processing; reference: diffusion.py:1466-1848)
# ------------------------------------------------------------------------------------------------
class PopulationModelDiffusion(nn.Module):
    def __init__(self, model=None, sde=None, shift=None, scale=None, method="dopri5", no_sigma=False,
                 hutchinson=False, options=None, *, precision="f32"):
        """Arguments as in the reference (diffusion.py:1466-1530); ``precision`` (keyword only, extension) is handed to the
        inner ``ScoreModel`` (see there)."""
        super().__init__()
        self.model = model
        self.sde = sde
        self.score_model = ScoreModel(model=self.model, sde=self.sde, hutchinson=hutchinson, no_sigma=no_sigma, precision=precision)
        n = self.model.n_dimensions
        self.register_buffer("shift", shift if shift is not None else torch.zeros(n, dtype=torch.float32))
        self.register_buffer("scale", scale if scale is not None else torch.ones(n, dtype=torch.float32))
        self.method = method
        self.options = options

    def forward(self, base_samples):
        if self.scale.dtype != torch.float32 or self.shift.dtype != torch.float32:
            # (e.g. float64 statistics from numpy: the reference's `* scale + shift` then promotes the result -- so does this)
            return self.score_model._sample_ode(base_samples, None, 1e-5, 1e-5, self.method, self.options) * self.scale + self.shift
        return self.score_model._sample_ode(base_samples, None, 1e-5, 1e-5, self.method, self.options,
                                            out_scale=self.scale, out_shift=self.shift)

    def sample_sde(self, shape, steps=100):
        # the reference ignores `steps` here and always takes 100 (diffusion.py:1608)
        return self.score_model.sample_sde(shape, steps=100) * self.scale + self.shift

    def log_prob(self, x, atol=1e-5, rtol=1e-5):
        # the reference does not forward self.method here: the solver default applies (diffusion.py:1633-1635)
        xT, lp = self.score_model._solve_forward(x, None, atol, rtol, "dopri5", self.options,
                                                 in_shift=self.shift, in_scale=self.scale)
        return lp + torch.sum(self.sde.prior(xT.shape).log_prob(xT), 1, keepdim=True)


class PopulationModelDiffusionConditional(nn.Module):
    def __init__(self, model=None, sde=None, shift=None, scale=None, conditional_shift=None,
                 conditional_scale=None, no_sigma=False, method="dopri5", options=None, *, precision="f32"):
        super().__init__()
        self.model = model
        self.sde = sde
        self.score_model = ScoreModel(model=self.model, sde=self.sde, no_sigma=no_sigma, precision=precision)
        n, c = self.model.n_dimensions, self.model.n_conditionals
        self.register_buffer("shift", shift if shift is not None else torch.zeros(n, dtype=torch.float32))
        self.register_buffer("scale", scale if scale is not None else torch.ones(n, dtype=torch.float32))
        self.register_buffer("conditional_shift",
                             conditional_shift if conditional_shift is not None else torch.zeros(c, dtype=torch.float32))
        self.register_buffer("conditional_scale",
                             conditional_scale if conditional_scale is not None else torch.ones(c, dtype=torch.float32))
        self.options = options
        self.method = method

    def _cond(self, conditional):
        return (conditional - self.conditional_shift) / self.conditional_scale

    def forward(self, base_samples, conditional=None):
        if self.scale.dtype != torch.float32 or self.shift.dtype != torch.float32:
            return self.score_model._sample_ode(base_samples, self._cond(conditional), 1e-5, 1e-5, self.method,
                                                self.options) * self.scale + self.shift
        return self.score_model._sample_ode(base_samples, self._cond(conditional), 1e-5, 1e-5, self.method,
                                            self.options, out_scale=self.scale, out_shift=self.shift)

    def sample_sde(self, shape, conditional=None, steps=100):
        return self.score_model.sample_sde(shape, conditional=self._cond(conditional), steps=100) * self.scale + self.shift

    def log_prob(self, x, conditional=None, atol=1e-5, rtol=1e-5):
        xT, lp = self.score_model._solve_forward(x, self._cond(conditional), atol, rtol, "dopri5", self.options,
                                                 in_shift=self.shift, in_scale=self.scale)
        return lp + torch.sum(self.sde.prior(xT.shape).log_prob(xT), 1, keepdim=True)
