"""Flow-matching continuous normalising flows on MI355X: the reference's ``flowfusion.flow`` API
(``ODEFlow`` flow.py:9-438, ``ConditionalODEFlow`` flow.py:441-941) with ``sample`` and
``log_prob`` / ``solve_ode_forward`` running as one fused HIP launch.

The velocity network takes ``[x, t]`` (``[x, t, (cond - shift)/scale]`` when conditional,
flow.py:112-115, 583-586).  ``state_dict`` keys match the reference: ``twopi``,
``target_shift``/``target_scale`` (+ ``conditional_*``), and the Linear parameters under both
``layers.{0,2,..}`` and ``velocity.{0,2,..}`` (the same modules registered twice).

Native: ``sample`` (t: 1 -> 0) and ``solve_ode_forward`` / ``log_prob`` (t: 0 -> 1, exact
divergence by forward-mode tangents, or a Hutchinson probe as an opt-in extension) for SiLU
networks and fixed-grid methods.  The reference's ``sample`` exposes no solver arguments and
always runs adaptive dopri5 at torchdiffeq's default tolerances (flow.py:299-303); here ``sample``
takes optional ``method``/``options`` keywords.  Nothing on these methods falls back to eager
PyTorch or the CPU: unsupported requests raise.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import nn

from . import _native, adaptive, device_adaptive, generic, solvers
from .fused import FusedNet, MODE_EXACT, MODE_HUTCH, MODE_STATE, activation_spec, require_fp32, within_envelope

_DEFAULT_SAMPLE_METHOD = "dopri5"     # what odeint() picks when the reference passes no method


def _build_layers(sizes, activation):
    layers = nn.ModuleList()
    for n_in, n_out in zip(sizes[:-2], sizes[1:-1]):
        layers.append(nn.Linear(n_in, n_out))
        layers.append(activation())
    layers.append(nn.Linear(sizes[-2], sizes[-1]))
    return layers


class _FlowBase(nn.Module):
    """Common fused plumbing of the two flow classes."""

    def _linears(self):
        return [m for m in self.layers if isinstance(m, nn.Linear)]

    def _net(self) -> FusedNet:
        specs = {activation_spec(m) for m in self.layers if not isinstance(m, nn.Linear)}
        if len(specs) != 1:
            raise NotImplementedError("the fused gfx950 path needs one activation shared by all hidden layers")
        act = next(iter(specs))
        lin = self._linears()
        cached = getattr(self, "_fused", None)
        prec = getattr(self, "precision", "f32")       # extension: "bf16x3" = split-precision kernels (see ScoreModel)
        if cached is None or not cached.serves(lin, act, prec):
            D = self.target_dimension
            C = getattr(self, "conditional_dimension", 0)
            # first-layer columns: [x (D) | t (1) | cond (C)]
            object.__setattr__(self, "_fused", FusedNet(lin, D, C, x_col0=0, c_col0=D + 1, act=act, precision=prec))
        return self._fused

    def _fusable(self) -> bool:
        """A compiled kernel holds the velocity network (shape and activation).  The reference has no limit on
        either (flow.py:61-74, 494-508); outside the compiled shapes the solve goes through generic.py.  An explicit
        ``precision=`` other than f32 never switches arithmetic silently: ``_net()`` raises instead."""
        if getattr(self, "precision", "f32") != "f32":
            return True
        key = tuple(id(m) for m in self.layers) + tuple(repr(m) for m in self.layers if not isinstance(m, nn.Linear))
        return within_envelope(self, key, self._net)

    def _solve_generic(self, x, t_span, method, options, mode, atol, rtol, cond, probe, raw_cond, affine):
        """The same solve around ``self.velocity`` evaluated by torch: stage combinations, error norms and step
        control by the library (generic.py), divergences by autograd as in the reference (flow.py:122-166, 598-652)."""
        D = self.target_dimension

        def velocity(t, y):
            cols = [y, t.reshape(1, 1).to(y.dtype).expand(y.shape[0], 1)]
            if cond is not None:
                cols.append(cond)
            return self.velocity(torch.cat(cols, dim=1))

        def rhs(t, y):
            if mode == MODE_STATE:
                with torch.no_grad():
                    return velocity(t, y), None
            with torch.enable_grad():
                y = y.detach().requires_grad_(True)
                v = velocity(t, y)
                if mode == MODE_HUTCH:
                    div = (torch.autograd.grad(v, y, probe)[0] * probe).sum(dim=1)
                else:
                    div = sum(torch.autograd.grad(v[:, i].sum(), y, retain_graph=True)[0][:, i] for i in range(D))
            return v.detach(), div.detach()

        if affine.get("in_shift") is not None:
            x = (x - affine["in_shift"]) / affine["in_scale"]
        extra = () if raw_cond is None else (raw_cond.detach().to(x.device, torch.float32),)
        y, lp, stats = generic.solve(rhs, x, t_span, method, options, mode != MODE_STATE, atol, rtol, norm_only=extra)
        self.last_solver_stats = stats
        if affine.get("out_scale") is not None:
            y = y * affine["out_scale"] + affine["out_shift"]
        return y, lp

    def _schedule(self, t, first=None):
        """(a, b, c1) for real times t (fp32, CPU): xdot = NET([x, t, cond]) -> a = 0, b = 1,
        c1 = w_t * t + bias of the first layer (flow.py:112-118).  ``first`` = host copy of the first layer,
        taken once per solve by the adaptive path (a device-to-host copy per step would wait for the kernel)."""
        w0, b0 = first if first is not None else self._net().first_layer_cpu()
        D = self.target_dimension
        with solvers.host_threads():
            c1 = t[:, None] * w0[:, D][None, :] + b0[None, :]
            return torch.zeros_like(t), torch.ones_like(t), c1

    def _table(self, t_span, method, options, mode, y0=None):
        plan = solvers.plan_ode(t_span, method, options, y0=y0)
        self._net().require_slots(int(plan.slot.max()) + 1, mode, f"method={method!r}")
        a, b, c1 = self._schedule(plan.t_eval)
        return solvers.build_table(plan, a, b, c1, self._net().width(mode))

    def _norm_cond(self, conditional):
        return (conditional - self.conditional_shift) / self.conditional_scale

    def _solve(self, x, t_span, method, options, mode, atol, rtol, cond=None, probe=None, raw_cond=None, **affine):
        require_fp32(self, x, cond, probe, what="an ODE solve")
        if not self._fusable():
            return self._solve_generic(x, t_span, method, options, mode, atol, rtol, cond, probe, raw_cond, affine)
        net = self._net()
        if method in solvers.ALL_ADAPTIVE:
            if any(v is not None for v in affine.values()):
                raise AssertionError("affine epilogues are applied by the caller on the adaptive path")
            t = t_span.double()
            sign = -1.0 if bool(t[0] > t[-1]) else 1.0
            # the reference keeps the raw conditional in the solver state (flow.py:779-796, 855-881)
            extra = () if raw_cond is None else (raw_cond.detach().to(x.device, torch.float32),)
            spec = None
            if x.is_cuda and method in solvers.NATIVE_ADAPTIVE:
                D = self.target_dimension
                w0t, b0 = net.time_columns(x.device, D, D + 1)
                spec = device_adaptive.ScheduleSpec(_native.SCHED_FLOW, (0.0, 0.0, 0.0), True, None, 0.0, w0t, b0)
            if device_adaptive.supported(spec, x, net, mode, options):
                # the whole loop on the device (device_adaptive.py): xdot = NET([x, t, cond]) -> a = 0, b = 1, c1 = w_t t + b1
                y, lp, stats = device_adaptive.solve(net, spec, sign, mode, x, float(sign * t[0]), float(sign * t[-1]), rtol, atol,
                                                     options, method, cond=cond, probe=probe, norm_only=extra)
                self.last_solver_stats = stats
                return y, lp
            first = net.first_layer_cpu()
            step = net.make_step(lambda tr: self._schedule(tr, first), sign, mode, x.device, cond=cond, probe=probe)
            solver = adaptive.make_solver(step, mode != MODE_STATE, rtol, atol, options, norm_only=extra, method=method, sign=sign)
            lp0 = torch.zeros(x.shape[0], device=x.device) if mode != MODE_STATE else None
            y, lp = solver.integrate(float(sign * t[0]), float(sign * t[-1]),
                                     x.detach().to(torch.float32).contiguous(), lp0)
            self.last_solver_stats = {"attempts": solver.n_attempts, "accepted": solver.n_accepted}
            return y, lp
        key = ("flow-ode", tuple(float(v) for v in t_span), method, repr(sorted((options or {}).items())), mode)
        if (options or {}).get("grid_constructor") is not None:      # the grid may depend on y0: built per call
            table = self._table(t_span, method, options, mode, y0=x).to(x.device)
        else:
            table = net.cached_table(key, x.device, lambda: self._table(t_span, method, options, mode))
        y, lp, _ = net.integrate(x, table, mode, cond=cond, probe=probe, stage_slots=solvers.resolve_method(method).stages, **affine)
        return y, (lp if mode != MODE_STATE else None)

    def _fused_sample(self, xT, conditional, method, options, atol, rtol, raw_cond=None):
        if torch.is_grad_enabled() and xT.requires_grad:
            raise NotImplementedError("gradients through the fused solve are not available; detach the input")
        method = _DEFAULT_SAMPLE_METHOD if method is None else method
        t_span = torch.tensor([1.0, 0.0], dtype=torch.float32)
        if method in solvers.ALL_ADAPTIVE:
            x, _ = self._solve(xT, t_span, method, options, MODE_STATE, atol, rtol, cond=conditional, raw_cond=raw_cond)
            return x * self.target_scale + self.target_shift
        x, _ = self._solve(xT, t_span, method, options, MODE_STATE, atol, rtol, cond=conditional,
                           out_scale=self.target_scale, out_shift=self.target_shift)
        return x

    def _probe_rng(self, probe, seed, sample_offset, hutchinson):
        """``probe="torch"`` (default): the +-1 probe is drawn on the CPU and moved; ``probe="philox"`` (keyword-only
        extension, as on ScoreModel.log_prob): the signs of the library's counter-based normals keyed by ``seed`` and the
        GLOBAL row ``sample_offset + r`` -- drawn on the device and independent of how a batch is cut into shards
        (``distributed.flow_log_prob_sharded``)."""
        if probe == "torch":
            if seed is not None:
                raise ValueError("seed= belongs to probe='philox' (the torch probe follows torch.manual_seed)")
            return None
        if probe != "philox":
            raise ValueError(f"probe must be 'torch' or 'philox', not {probe!r}")
        if not hutchinson:
            raise ValueError("probe='philox' is the Hutchinson probe: pass hutchinson=True")
        if seed is None:      # one draw of torch's generator, so torch.manual_seed still fixes the run
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        return int(seed), int(sample_offset)

    def _fused_forward(self, x, conditional, method, options, hutchinson, atol, rtol, raw_cond=None, probe_rng=None):
        t_span = torch.tensor([0.0, 1.0], dtype=torch.float32)
        mode, probe = MODE_EXACT, None
        if hutchinson and probe_rng is not None:
            mode = MODE_HUTCH
            z = _native.normal_fill(x.shape[0], x.shape[1], probe_rng[0], probe_rng[1], x.device,
                                    noise_index=_native.PROBE_NOISE_INDEX)
            probe = torch.where(z >= 0, 1.0, -1.0).to(torch.float32)
        elif hutchinson:
            mode = MODE_HUTCH
            probe = torch.sign(torch.randn(x.shape)).to(x.device)
        xT, logj = self._solve(x, t_span, method, options, mode, atol, rtol, cond=conditional, probe=probe,
                               raw_cond=raw_cond)
        return xT, logj.view(-1, 1)


class ODEFlow(_FlowBase):
    """Unconditional flow-matching CNF (reference: flow.py:9-438)."""

    def __init__(self, target_dimension: int = 1, hidden_units: List[int] = [128, 128],
                 activation: nn.Module = nn.SiLU, target_shift: Optional[torch.Tensor] = None,
                 target_scale: Optional[torch.Tensor] = None):
        super().__init__()
        self.target_dimension = target_dimension
        self.layers = _build_layers([target_dimension + 1] + list(hidden_units) + [target_dimension], activation)
        self.velocity = nn.Sequential(*self.layers)
        self.register_buffer("twopi", torch.tensor(2.0 * 3.14159265358979323846))
        self.register_buffer("target_shift", target_shift if target_shift is not None else torch.zeros(target_dimension))
        self.register_buffer("target_scale", target_scale if target_scale is not None else torch.ones(target_dimension))

    # -- pointwise (plain torch) ----------------------------------------------------------------
    def dynamics(self, t: torch.Tensor, states: Tuple[torch.Tensor]):
        x = states[0]
        return self.velocity(torch.cat([x, t.view(-1, 1).expand(x.shape[0], 1)], dim=1))

    def dynamics_with_jacobian(self, t, states):
        x, logj = states
        with torch.set_grad_enabled(True):
            x.requires_grad_(True)
            v = self.dynamics(t, (x,))
            div = torch.zeros_like(logj)
            for i in range(x.shape[-1]):
                div = div + torch.autograd.grad(v[:, i].sum(), x, create_graph=True, retain_graph=True)[0][:, i:i + 1]
        return v, div

    def forward(self, t, states):
        return self.dynamics(t, states)

    def compute_linear_velocity_field(self, x0, xT, t):
        x0 = (x0 - self.target_shift) / self.target_scale
        return (1 - t) * x0 + t * xT, xT - x0

    # -- fused ------------------------------------------------------------------------------------
    def sample(self, xT: torch.Tensor, gradients: bool = False, method: Optional[str] = None,
               options: Optional[dict] = None, atol: float = 1e-9, rtol: float = 1e-7):
        """Transport base samples xT (t=1) to the target (t=0), then ``* target_scale + target_shift``."""
        if gradients:
            raise NotImplementedError("sample(gradients=True) uses odeint_adjoint in the reference "
                                      "(flow.py:286-295); differentiable solves are out of scope")
        return self._fused_sample(xT, None, method, options, atol, rtol)

    def solve_ode_forward(self, x, atol: float = 1e-5, rtol: float = 1e-5, method: str = "dopri5",
                          options: Optional[dict] = None, adjoint: bool = False, hutchinson: bool = False, *,
                          probe: str = "torch", seed: Optional[int] = None, sample_offset: int = 0):
        """Integrate t: 0 -> 1 with the divergence; returns ``(xT, log_jacobian[B,1])`` (flow.py:308-384).
        ``probe`` / ``seed`` / ``sample_offset``: see ``_probe_rng``."""
        if adjoint:
            raise NotImplementedError("adjoint=True (odeint_adjoint) is out of scope for the fused path")
        return self._fused_forward(x, None, method, options, hutchinson, atol, rtol,
                                   probe_rng=self._probe_rng(probe, seed, sample_offset, hutchinson))

    def log_prob(self, x, atol: float = 1e-5, rtol: float = 1e-5, method: str = "dopri5",
                 options: Optional[dict] = None, adjoint: bool = False, hutchinson: bool = False, *,
                 probe: str = "torch", seed: Optional[int] = None, sample_offset: int = 0):
        """Log-density of target-space points, shape [B] (flow.py:386-438)."""
        x = (x - self.target_shift) / self.target_scale
        xT, logj = self.solve_ode_forward(x, atol, rtol, method, options, adjoint, hutchinson=hutchinson,
                                          probe=probe, seed=seed, sample_offset=sample_offset)
        base = torch.sum(-0.5 * xT ** 2 - 0.5 * torch.log(self.twopi), dim=1)
        return base + logj.squeeze(1) - torch.sum(torch.log(self.target_scale))


class ConditionalODEFlow(_FlowBase):
    """Conditional flow-matching CNF (reference: flow.py:441-941)."""

    def __init__(self, target_dimension: int = 1, conditional_dimension: int = 1,
                 hidden_units: List[int] = [128, 128], activation: nn.Module = nn.SiLU,
                 target_shift: Optional[torch.Tensor] = None, target_scale: Optional[torch.Tensor] = None,
                 conditional_shift: Optional[torch.Tensor] = None,
                 conditional_scale: Optional[torch.Tensor] = None):
        super().__init__()
        self.target_dimension = target_dimension
        self.conditional_dimension = conditional_dimension
        self.layers = _build_layers(
            [target_dimension + 1 + conditional_dimension] + list(hidden_units) + [target_dimension], activation)
        self.velocity = nn.Sequential(*self.layers)
        self.register_buffer("twopi", torch.tensor(2.0 * 3.14159265358979323846))
        self.register_buffer("target_shift", target_shift if target_shift is not None else torch.zeros(target_dimension))
        self.register_buffer("target_scale", target_scale if target_scale is not None else torch.ones(target_dimension))
        self.register_buffer("conditional_shift",
                             conditional_shift if conditional_shift is not None else torch.zeros(conditional_dimension))
        self.register_buffer("conditional_scale",
                             conditional_scale if conditional_scale is not None else torch.ones(conditional_dimension))

    # -- pointwise (plain torch) ----------------------------------------------------------------
    def dynamics(self, t, states):
        x, conditional = states
        c = self._norm_cond(conditional)
        v = self.velocity(torch.cat([x, t.view(-1, 1).expand(x.shape[0], 1), c], dim=1))
        return v, torch.zeros_like(c)

    def dynamics_with_jacobian(self, t, states):
        x, conditional, logj = states
        with torch.set_grad_enabled(True):
            x.requires_grad_(True)
            v = self.dynamics(t, (x, conditional))[0]
            div = torch.zeros_like(logj)
            for i in range(x.shape[-1]):
                div = div + torch.autograd.grad(v[:, i].sum(), x, create_graph=True, retain_graph=True)[0][:, i:i + 1]
        return v, torch.zeros_like(conditional), div

    def forward(self, t, states):
        return self.dynamics(t, states)

    def compute_linear_velocity_field(self, x0, xT, t):
        x0 = (x0 - self.target_shift) / self.target_scale
        return (1 - t) * x0 + t * xT, xT - x0

    # -- fused ------------------------------------------------------------------------------------
    def sample(self, xT, conditional, gradients: bool = False, method: Optional[str] = None,
               options: Optional[dict] = None, atol: float = 1e-9, rtol: float = 1e-7):
        if gradients:
            raise NotImplementedError("sample(gradients=True) uses odeint_adjoint in the reference "
                                      "(flow.py:779-788); differentiable solves are out of scope")
        return self._fused_sample(xT, self._norm_cond(conditional), method, options, atol, rtol, raw_cond=conditional)

    def solve_ode_forward(self, x, conditional, atol: float = 1e-5, rtol: float = 1e-5,
                          method: str = "dopri5", options: Optional[dict] = None, adjoint: bool = False,
                          hutchinson: bool = False, *, probe: str = "torch", seed: Optional[int] = None,
                          sample_offset: int = 0):
        if adjoint:
            raise NotImplementedError("adjoint=True (odeint_adjoint) is out of scope for the fused path")
        return self._fused_forward(x, self._norm_cond(conditional), method, options, hutchinson, atol, rtol,
                                   raw_cond=conditional, probe_rng=self._probe_rng(probe, seed, sample_offset, hutchinson))

    def log_prob(self, x, conditional, atol: float = 1e-5, rtol: float = 1e-5, method: str = "dopri5",
                 options: Optional[dict] = None, adjoint: bool = False, hutchinson: bool = False, *,
                 probe: str = "torch", seed: Optional[int] = None, sample_offset: int = 0):
        x = (x - self.target_shift) / self.target_scale
        xT, logj = self.solve_ode_forward(x, conditional, atol, rtol, method, options, adjoint,
                                          hutchinson=hutchinson, probe=probe, seed=seed, sample_offset=sample_offset)
        base = torch.sum(-0.5 * xT ** 2 - 0.5 * torch.log(self.twopi), dim=1)
        return base + logj.squeeze(1) - torch.sum(torch.log(self.target_scale))
