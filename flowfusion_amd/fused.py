"""Shared host plumbing between the score-model and flow front ends.

``FusedNet`` owns what the kernel needs from a Linear/activation stack: the kernel plan, the
weights repacked into MFMA operand order (cached on the device, refreshed when a parameter
changes) and the launch itself.  The front ends (diffusion.py, flow.py) supply the
time-dependent part: one evaluation table per solve (solvers.py).
"""
from __future__ import annotations

import warnings
from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import _native
from ._native import MODE_EXACT, MODE_HUTCH, MODE_STATE  # noqa: F401  (re-exported)


def activation_spec(act) -> Tuple[int, float, float]:
    """(FF_ACT_* code, parameter 0, parameter 1) of a torch.nn activation module or class -- the
    `activation` argument of the reference constructors (an instance for MLP, diffusion.py:38; a class
    for the flows, flow.py:41,70)."""
    if isinstance(act, type):
        act = act()
    N = _native
    if type(act) is nn.SiLU:
        return (N.ACT_SILU, 0.0, 0.0)
    if type(act) is nn.Tanh:
        return (N.ACT_TANH, 0.0, 0.0)
    if type(act) is nn.Sigmoid:
        return (N.ACT_SIGMOID, 0.0, 0.0)
    if type(act) is nn.ReLU:
        return (N.ACT_RELU, 0.0, 0.0)
    if type(act) is nn.LeakyReLU:
        return (N.ACT_LEAKY_RELU, float(act.negative_slope), 0.0)
    if type(act) is nn.ELU:
        return (N.ACT_ELU, float(act.alpha), 0.0)
    if type(act) is nn.Softplus:
        return (N.ACT_SOFTPLUS, float(act.beta), float(act.threshold))
    if type(act) is nn.GELU:
        return (N.ACT_GELU_TANH if act.approximate == "tanh" else N.ACT_GELU, 0.0, 0.0)
    raise NotImplementedError(
        f"activation {act!r}: the fused gfx950 kernels implement SiLU (the reference default), Tanh, Sigmoid, "
        "ReLU, LeakyReLU, ELU, Softplus and GELU")


class FusedEnvelopeWarning(UserWarning):
    """A network no compiled kernel can hold is evaluated by torch on the GPU (stepping stays native)."""


def within_envelope(owner, key, build: Callable[[], "FusedNet"]) -> bool:
    """True if a compiled kernel holds the network ``build()`` describes (plans for the state-only and the
    divergence-capable kernels exist).  The reference puts no limit on width, dimension, conditional inputs or
    activation (flowfusion/diffusion.py:59-72, flow.py:61-74); outside the compiled shapes the front ends keep the
    solve on the GPU by handing the network to ``generic.py`` -- the module evaluated by torch, every Runge-Kutta
    combination, error norm and noise update by the library's kernels -- and say so once per network
    (``FusedEnvelopeWarning``).  The answer is cached on ``owner`` under ``key`` (identity of the layers)."""
    cached = owner.__dict__.get("_envelope")
    if cached is not None and cached[0] == key:
        return cached[1]
    try:
        net = build()
        net.plan(MODE_STATE)
        net.plan(MODE_EXACT)
        ok = True
    except NotImplementedError as e:
        warnings.warn(f"{e} -- outside the fused kernels' envelope: this network is evaluated by torch on the GPU, "
                      "time stepping stays in the library's kernels (flowfusion_amd.generic)",
                      FusedEnvelopeWarning, stacklevel=4)
        ok = False
    object.__setattr__(owner, "_envelope", (key, ok))
    return ok


def exact_trace_passes(dim: int, tile: int) -> List[Tuple[int, int]]:
    """Split the `dim` unit tangents of an exact trace into launches [(first, count), ...].

    A wavefront carries `tile` MFMA columns; a launch with n tangents per sample packs
    floor(tile / (1 + n)) samples into them, so it costs tile / floor(tile / (1 + n)) columns per
    sample (the value column is recomputed by every launch).  The cheapest partition is not always
    the fewest launches: 8 dimensions on 16 columns cost 16 in one launch (9 columns, one sample per
    wavefront) but 8 + 2 as 7 + 1.  Small dynamic programme over the partitions of `dim`."""
    cost = [0.0] + [tile / (tile // (1 + n)) for n in range(1, tile)]
    best = [(0.0, [])] + [None] * dim
    for d in range(1, dim + 1):
        cands = []
        for n in range(1, min(d, tile - 1) + 1):
            c, parts = best[d - n]
            cands.append((c + cost[n] + 1e-9 * (len(parts) + 1), parts + [n]))   # ties: fewer launches
        best[d] = min(cands, key=lambda cp: cp[0])
    out, first = [], 0
    for n in sorted(best[dim][1], reverse=True):
        out.append((first, n))
        first += n
    return out


class FusedNet:
    """Kernel-side view of ``Linear -> SiLU -> ... -> Linear`` with first-layer input
    ``[ time part | x | cond ]`` in any column order."""

    def __init__(self, linears: Sequence[nn.Linear], dim: int, cond_dim: int, x_col0: int, c_col0: int,
                 act: Tuple[int, float, float] = (_native.ACT_SILU, 0.0, 0.0), precision: str = "f32"):
        self.linears = list(linears)
        self.act = (int(act[0]), float(act[1]), float(act[2]))
        if precision not in _native.PRECISIONS:
            raise ValueError(f"precision={precision!r}: expected one of {sorted(_native.PRECISIONS)}")
        self.precision = precision
        if len(self.linears) < 2:
            raise NotImplementedError("the fused path needs at least one hidden layer")
        self.dim = int(dim)
        self.cond_dim = int(cond_dim)
        self.x_col0 = int(x_col0)
        self.c_col0 = int(c_col0)
        self.hidden = [int(l.out_features) for l in self.linears[:-1]]
        if int(self.linears[-1].out_features) != self.dim:
            raise ValueError("last layer must produce `dim` outputs")
        for l in self.linears:
            if l.bias is None:
                raise NotImplementedError("Linear layers without bias are not supported")
        self._plans = {}
        self._wpack = {}            # layout key -> (parameter versions, packed device tensor)
        self._tables = {}           # small cache of evaluation tables already on the device

    def serves(self, linears, act, precision: str = "f32") -> bool:
        """True if this view was built from exactly these Linear modules (identity, in order) and this
        activation.  The front ends rebuild the view otherwise: a layer replaced after the first solve
        (``model.NN[2] = nn.Linear(..)``) must not keep integrating with the old weights."""
        act = (int(act[0]), float(act[1]), float(act[2]))
        return (self.act == act and self.precision == precision and len(linears) == len(self.linears)
                and all(a is b for a, b in zip(linears, self.linears)))

    # -- plan / weights ---------------------------------------------------------------------
    def plan(self, mode: int) -> _native.PlanStruct:
        key = 0 if mode == MODE_STATE else (1 if mode == MODE_HUTCH else 2)
        if key not in self._plans:
            self._plans[key] = _native.make_plan(self.dim, self.cond_dim, self.hidden, mode, self.act,
                                                 _native.PRECISIONS[self.precision])
        return self._plans[key]

    def _param_key(self, device, plan) -> Tuple:
        """(layout key, parameter versions).  The packed layout is a function of (tile, width, dregs, cregs,
        n_hidden) -- ff_layout.h make_layout -- and not of the kernel instantiation: the state-only and the
        divergence-capable kernels of one shape share a buffer."""
        vers = tuple((p.data_ptr(), p._version) for l in self.linears for p in (l.weight, l.bias))
        return (str(device), plan.precision, plan.tile, plan.width, plan.dregs, plan.cregs, plan.n_hidden), vers

    def wpack(self, device, mode: int) -> torch.Tensor:
        plan = self.plan(mode)
        layout, vers = self._param_key(device, plan)
        hit = self._wpack.get(layout)
        if hit is None or hit[0] != vers:
            packed = _native.pack_weights(
                plan, [l.weight for l in self.linears], [l.bias for l in self.linears],
                self.hidden, self.x_col0, self.c_col0)
            hit = (vers, packed.to(device))
            self._wpack[layout] = hit
        return hit[1]

    # -- launch -----------------------------------------------------------------------------
    def integrate(self, x: torch.Tensor, etab: torch.Tensor, mode: int = MODE_STATE,
                  cond: Optional[torch.Tensor] = None, probe: Optional[torch.Tensor] = None,
                  noise: Optional[torch.Tensor] = None,
                  in_shift=None, in_scale=None, out_scale=None, out_shift=None,
                  rng: Optional[Tuple[int, int, int]] = None, stage_slots: int = 0):
        """Run the fused integration.  Returns (y_final [B,D], dlogp [B] or empty, status [1]).
        ``rng = (seed, global index of row 0, noise index of table row 0)`` selects in-kernel noise for
        tables with noise rows (instead of a ``noise`` buffer)."""
        if not x.is_cuda:
            raise RuntimeError(
                "flowfusion_amd integrates on the GPU only: move the model and its inputs to 'cuda' "
                f"(got a tensor on {x.device}); there is no CPU fallback")
        if x.dim() != 2 or x.shape[1] != self.dim:
            raise ValueError(f"expected a [batch, {self.dim}] state, got {tuple(x.shape)}")
        dev = x.device
        plan = self.plan(mode)
        f32 = lambda t: None if t is None else t.detach().to(dev, torch.float32).contiguous()
        if self.cond_dim > 0:
            if cond is None:
                raise ValueError("this network has conditional inputs; `conditional` is required")
            cond = f32(cond)
            if cond.dim() != 2 or cond.shape != (x.shape[0], self.cond_dim):
                raise ValueError(f"conditional must be [batch, {self.cond_dim}], got {tuple(cond.shape)}")
        else:
            cond = None
        args = (f32(x), cond, f32(probe), f32(noise), self.wpack(dev, mode), f32(etab),
                f32(in_shift), f32(in_scale), f32(out_scale), f32(out_shift), _native.plan_words(plan, stage_slots), mode)
        if rng is not None:
            if mode != MODE_STATE or noise is not None:
                raise ValueError("in-kernel noise applies to state-only integration without a noise buffer")
            return torch.ops.flowfusion_amd.mlp_ode(*args, 0, 0, int(rng[0]), int(rng[1]), int(rng[2]))
        if mode != MODE_EXACT:
            return torch.ops.flowfusion_amd.mlp_ode(*args)
        # exact trace = sum over dimensions of unit-tangent contributions: integrate it in the
        # cheapest set of launches (each carries at most tile - 1 tangents per sample) and add up
        total = status = None
        for first, count in exact_trace_passes(self.dim, plan.tile):
            y, dl, st = torch.ops.flowfusion_amd.mlp_ode(*args, first, count)
            total = dl if total is None else total + dl
            status = st if status is None else (status | st)
        return y, total, status

    def cached_table(self, key, device, build):
        """Evaluation tables depend only on (time span, method, options, schedule parameters, first-layer
        weights): keep the last few on the device so repeated solves skip the host work."""
        l0 = self.linears[0]
        full = (key, str(device), l0.weight.data_ptr(), l0.weight._version, l0.bias._version)
        hit = self._tables.get(full)
        if hit is None:
            if len(self._tables) >= 8:
                self._tables.pop(next(iter(self._tables)))
            hit = build().to(device)
            self._tables[full] = hit
        return hit

    # -- adaptive stepping under the HOST controller: one launch per attempted step (flowfusion_amd/adaptive.py) -----------
    def make_step(self, schedule, sign: float, mode: int, device, cond=None, probe=None, launcher=None):
        """Step function for ``adaptive.Dopri5``.  ``schedule(t_real fp32 [n]) -> (a, b, c1)`` supplies
        the time-dependent scalars and first-layer bias; ``sign`` = -1 for a decreasing span (solved
        in reversed time with the right-hand side negated).  ``launcher`` replaces the GPU launch in
        the CPU tests (kernel-semantics emulator)."""
        plan = self.plan(mode)
        width = plan.width
        f32 = lambda t: None if t is None else t.detach().to(device, torch.float32).contiguous()
        cond_d = f32(cond) if self.cond_dim > 0 else None
        probe_d = f32(probe)
        own = launcher is None
        if own:
            wpack = self.wpack(device, mode)
            launcher = lambda y, k1, kl1, lp0, etab, n_aux, first, count, used=0: torch.ops.flowfusion_amd.mlp_ode_step(
                y, cond_d, probe_d, k1, kl1, lp0, wpack, etab.to(device), _native.plan_words(plan, used), mode, n_aux, first, count)
        passes = [(0, 0)] if mode != MODE_EXACT else exact_trace_passes(self.dim, plan.tile)

        def step(y, k1, lp0, kl1, t_rows, cin, slots, tail, use_y, n_aux):
            n = int(t_rows.numel())
            used = int(slots.max()) + 1
            self.require_slots(used, mode, "this adaptive method")
            hint = {"used": used} if own else {}
            a, b, c1 = schedule(sign * t_rows)
            rows = torch.zeros(n + 2, 32 + width, dtype=torch.float32)
            rows[:n, 0] = sign * a
            rows[:n, 1] = sign * b
            ints = rows.view(torch.int32)
            ints[:n, 4] = slots
            rows[:n, 8:16] = cin
            rows[:n, 32:32 + c1.shape[1]] = c1
            rows[n, 8:16], rows[n, 16:24] = tail[0], tail[1]
            rows[n + 1, 8:16], rows[n + 1, 16:24] = tail[2], tail[3]
            ints[n, 3] = use_y
            aux = aux_lp = None
            for i, (first, count) in enumerate(passes):
                o, olp = launcher(y, k1, kl1 if i == 0 else None, lp0 if i == 0 else None, rows, n_aux, first, count, **hint)
                aux = o if aux is None else aux
                aux_lp = olp if aux_lp is None else aux_lp + olp
            return aux, (aux_lp if mode != MODE_STATE else None)

        return step

    # -- first layer pieces used by the table builders ---------------------------------------
    def time_columns(self, device, c0: int, c1: int):
        """(W1[:, c0:c1] contiguous, bias) of the first layer as fp32 device tensors -- what the device-side adaptive
        controller multiplies the time features with (ff_adapt_config.w0t / b0).  Cached until a parameter changes."""
        l0 = self.linears[0]
        key = (str(device), c0, c1, l0.weight.data_ptr(), l0.weight._version, l0.bias.data_ptr(), l0.bias._version)
        hit = self.__dict__.get("_time_cols")
        if hit is None or hit[0] != key:
            w = l0.weight.detach()[:, c0:c1].to(device, torch.float32).contiguous()
            b = l0.bias.detach().to(device, torch.float32).contiguous()
            hit = (key, w, b)
            self._time_cols = hit
        return hit[1], hit[2]

    def first_layer_cpu(self):
        l0 = self.linears[0]
        return (l0.weight.detach().to("cpu", torch.float32), l0.bias.detach().to("cpu", torch.float32))

    def width(self, mode: int = MODE_STATE) -> int:
        return int(self.plan(mode).width)

    def stage_slots(self, mode: int = MODE_STATE) -> int:
        """Runge-Kutta stage slots the selected kernel keeps on chip: 7 (FF_MAX_SLOTS) everywhere except the split-precision
        kernels for states of 17-32 dimensions, which keep 4 (csrc/ff_split_layout.h slots_on_chip)."""
        plan = self.plan(mode)
        return 4 if (plan.precision != _native.PREC_F32 and plan.dregs == 16) else 7

    def require_slots(self, used: int, mode: int, what: str):
        if used > self.stage_slots(mode):
            raise NotImplementedError(
                f"precision={self.precision!r} with {self.dim} state dimensions keeps {self.stage_slots(mode)} Runge-Kutta stage "
                f"slots on chip and {what} needs {used}: use euler / midpoint / heun3 / rk4 / rk4_classic (or bosh3 / fehlberg2 "
                "/ adaptive_heun), or precision='f32'")


def require_fp32(module, *tensors, what="this solve"):
    """The kernels compute in fp32, the dtype the reference's constructors create (and every BASELINE configuration uses).
    The reference itself follows the dtype of its parameters and inputs -- a ``.double()`` model solves in float64 there --
    so anything else is refused here rather than rounded to fp32 behind the caller's back."""
    for p in module.parameters():
        if p.dtype != torch.float32:
            raise TypeError(f"flowfusion_amd: {what} computes in float32; the model holds {p.dtype} parameters (the reference "
                            "would solve in that dtype) -- cast the model with .float()")
    for t in tensors:
        if t is not None and torch.is_tensor(t) and t.is_floating_point() and t.dtype != torch.float32:
            raise TypeError(f"flowfusion_amd: {what} computes in float32; got a {t.dtype} input (the reference would "
                            "solve in that dtype) -- cast it with .float()")
