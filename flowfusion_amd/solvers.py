"""Host side of the fused integrator: time grids, Runge-Kutta tableaux, evaluation tables.

The reference never steps an ODE itself: every solve is ``torchdiffeq.odeint(func, y0, t,
method=, options=)`` (flowfusion/diffusion.py:631-639, 744-752; flowfusion/flow.py:299-303,
371-382, 792-796, 869-881) and the reverse SDE is a Python loop (diffusion.py:539-559).
Everything in those loops that depends only on *time* -- the grid, the stage times, the
SDE schedule scalars, the time embedding pushed through the first layer -- is identical
for every sample, so it is computed here once, in fp32 torch arithmetic that follows the
reference's operation order, and handed to the kernel as one row per RHS evaluation
(layout: ``ff::RowHdr`` in csrc/ff_layout.h followed by the first-layer bias ``c1``).

torchdiffeq (pinned >=0.2.5,<0.3.0 by the reference's pyproject.toml:12) is not available
offline; the fixed-grid semantics below restate its published algorithm: the grid is
``arange(ceil((t1-t0)/step_size + 1)) * step_size + t0`` with the last point clamped to
``t1`` (or just ``t`` when no ``step_size`` is given); a decreasing time span is solved as
``-t`` with the right-hand side negated; ``rk4`` is the 3/8-rule; the tuple state is one
flat vector.  Parity of these rules with the real package is unpinned.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional, Sequence

import torch

class host_threads:
    """Context for the small host-side computations of this package (schedules, tables): cap torch's intra-op
    threads.  On a many-core host the default pool (e.g. 128 threads) takes milliseconds to wake for operands of
    a few thousand elements -- longer than the kernels these tables feed at notebook batch sizes."""

    def __init__(self, n: int = 4):
        self.n = n

    def __enter__(self):
        self.prev = torch.get_num_threads()
        if self.prev > self.n:
            torch.set_num_threads(self.n)
        return self

    def __exit__(self, *exc):
        if torch.get_num_threads() != self.prev:
            torch.set_num_threads(self.prev)
        return False


ROW_HDR = 32          # words in the row header (FF_ROW_HDR)
MAX_SLOTS = 7         # stage slots on chip (FF_MAX_SLOTS)
FLAG_STEP_END = 1
FLAG_NOISE = 2

_one_third = 1.0 / 3.0
_two_thirds = 2.0 / 3.0


@dataclass(frozen=True)
class Tableau:
    """Explicit Runge-Kutta scheme: stage i is evaluated at ``t0 + c[i]*dt`` (``None`` means
    "exactly t1") on ``y0 + dt * sum_j A[i][j] k_j``; the step is ``dt * sum_j b[j] k_j``."""
    name: str
    c: Sequence[Optional[float]]
    A: Sequence[Sequence[float]]
    b: Sequence[float]

    @property
    def stages(self) -> int:
        return len(self.b)


# torchdiffeq names (fixed-grid family) ----------------------------------------------------------
EULER = Tableau("euler", (0.0,), ((),), (1.0,))
MIDPOINT = Tableau("midpoint", (0.0, 0.5), ((), (0.5,)), (0.0, 1.0))
HEUN3 = Tableau("heun3", (0.0, _one_third, _two_thirds),
                ((), (_one_third,), (0.0, _two_thirds)), (0.25, 0.0, 0.75))
# torchdiffeq's "rk4" is the 3/8 rule (rk4_alt_step_func); its last stage is taken at t1 itself.
RK4_38 = Tableau("rk4", (0.0, _one_third, _two_thirds, None),
                 ((), (_one_third,), (-_one_third, 1.0), (1.0, -1.0, 1.0)),
                 (0.125, 0.375, 0.375, 0.125))
# extensions (not torchdiffeq names) ---------------------------------------------------------------
RK4_CLASSIC = Tableau("rk4_classic", (0.0, 0.5, 0.5, None),
                      ((), (0.5,), (0.0, 0.5), (0.0, 0.0, 1.0)),
                      (1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0))
# Dormand-Prince 5(4) taken at a fixed step (5th-order weights; the 7th FSAL stage only feeds the
# error estimate of the adaptive method and is not needed here): 6 evaluations per step.
DOPRI5_FIXED = Tableau(
    "dopri5_fixed",
    (0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, None),
    ((),
     (1 / 5,),
     (3 / 40, 9 / 40),
     (44 / 45, -56 / 15, 32 / 9),
     (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
     (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656)),
    (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84),
)

FIXED_METHODS = {t.name: t for t in (EULER, MIDPOINT, HEUN3, RK4_38, RK4_CLASSIC, DOPRI5_FIXED)}
ADAPTIVE_METHODS = ("dopri5", "dopri8", "bosh3", "fehlberg2", "adaptive_heun")
NATIVE_ADAPTIVE = ("dopri5", "bosh3", "fehlberg2", "adaptive_heun")     # adaptive.TABLEAUX: one fused launch per attempted step
HOST_STEPPED_ADAPTIVE = ("dopri8",)    # adaptive.WIDE_TABLEAUX: 13 stages > 7 slots, one fused launch per STAGE
ALL_ADAPTIVE = NATIVE_ADAPTIVE + HOST_STEPPED_ADAPTIVE


def resolve_method(method: str) -> Tableau:
    if method in FIXED_METHODS:
        return FIXED_METHODS[method]
    if method in ADAPTIVE_METHODS:
        raise NotImplementedError(f"method={method!r} is adaptive: no fixed-grid tableau (the front ends route it to the adaptive drivers)")
    if method in ("explicit_adams", "implicit_adams", "fixed_adams", "scipy_solver"):
        raise NotImplementedError(
            f"method={method!r}: torchdiffeq's multistep / scipy solvers are not built (adaptive: {ALL_ADAPTIVE}; fixed grids: "
            f"{sorted(FIXED_METHODS)} with options={{'step_size': h}})")
    raise ValueError(f"unknown ODE method {method!r}; supported: {sorted(FIXED_METHODS)} and {ALL_ADAPTIVE}")


def fixed_grid(t: torch.Tensor, step_size: Optional[float]) -> torch.Tensor:
    """Solver-time grid for the (ascending) evaluation times ``t`` -- fp32, as torchdiffeq builds it."""
    if step_size is None:
        return t.clone()
    start, end = t[0], t[-1]
    niters = int(torch.ceil((end - start) / step_size + 1).item())
    grid = torch.arange(0, niters, dtype=t.dtype) * step_size + start
    grid[-1] = t[-1]
    return grid


@dataclass
class EvalPlan:
    """Evaluation times (real time, fp32) plus the per-row integrator words."""
    t_eval: torch.Tensor        # [n_evals] real (un-reversed) time of each RHS evaluation
    sign: float                 # +1 forward, -1 if the span is decreasing (RHS negated, solved in -t)
    slot: torch.Tensor          # [n_evals] int32
    flags: torch.Tensor         # [n_evals] int32
    cin: torch.Tensor           # [n_evals, 8]
    cout: torch.Tensor          # [n_evals, 8]
    n_steps: int


def plan_ode(t_span: torch.Tensor, method: str, options: Optional[dict], y0=None) -> EvalPlan:
    """Rows for ``odeint(func, y0, t_span, method=method, options=options)`` on a fixed grid.

    ``t_span`` is the fp32 two-point tensor the reference builds (e.g. ``[1.0, epsilon]``,
    diffusion.py:611; ``[epsilon, 1.0]`` :727; ``[1.0, 0.0]`` flow.py:282; ``[0.0, 1.0]`` :354).
    ``y0`` (optional) is handed to a ``grid_constructor``.
    """
    tab = resolve_method(method)
    if tab.stages > MAX_SLOTS:
        raise NotImplementedError(f"{method}: {tab.stages} stages exceed the {MAX_SLOTS} on-chip slots")
    options = dict(options or {})
    step_size = options.pop("step_size", None)
    options.pop("min_step", None)      # adaptive-only knob (the reference's log_prob default passes it)
    # torchdiffeq's other fixed-grid options (FixedGridODESolver.__init__): `grid_constructor(func, y0, t)` returns the
    # time grid instead of `step_size`.  On a decreasing span torchdiffeq's `_check_inputs` wraps the user's function as
    # `lambda func, y0, t: -grid_constructor(func, y0, -t)`: the constructor sees and returns REAL (decreasing) times,
    # the solver works on the negated grid.  `func` is not known where the table is built and is passed as None.
    # `perturb=True` evaluates a step's first stage at the next
    # representable time after t0 and a stage taken at t1 itself at the one before t1 (right-hand sides with jumps on
    # the grid), `interp` only matters for output times inside a step -- the reference asks for the end points only
    grid_constructor = options.pop("grid_constructor", None)
    perturb = bool(options.pop("perturb", False))
    interp = options.pop("interp", "linear")
    if interp not in ("linear", "cubic"):
        raise ValueError(f"Unknown interpolation method {interp}")
    t_span = t_span.detach().to("cpu", torch.float32)
    sign = 1.0
    if bool(t_span[0] > t_span[-1]):
        sign = -1.0
        t_span = -t_span            # torchdiffeq solves decreasing spans in reversed time
    if grid_constructor is not None:
        if step_size is not None:
            raise ValueError("step_size and grid_constructor are mutually exclusive arguments.")
        grid = torch.as_tensor(grid_constructor(None, y0, sign * t_span)).detach().to("cpu", torch.float32).reshape(-1)
        grid = sign * grid          # solver time
        if grid.numel() < 2 or grid[0] != t_span[0] or grid[-1] != t_span[-1]:
            raise AssertionError("grid_constructor must return a grid that starts at t[0] and ends at t[-1]")
        if not bool((grid[1:] > grid[:-1]).all()):
            raise ValueError("grid_constructor must return times strictly monotonic in the direction of the span")
    else:
        grid = fixed_grid(t_span, step_size)
    if grid.numel() < 2:
        raise ValueError("time grid has fewer than two points")
    t0s, t1s = grid[:-1], grid[1:]
    dts = t1s - t0s
    n_steps = int(dts.numel())
    S = tab.stages
    n_evals = n_steps * S
    tau = torch.empty(n_steps, S, dtype=torch.float32)
    for i, c in enumerate(tab.c):
        if c is None:
            tau[:, i] = torch.nextafter(t1s, t1s - 1) if perturb else t1s
        elif c == 0.0:
            tau[:, i] = torch.nextafter(t0s, t0s + 1) if perturb else t0s
        else:
            tau[:, i] = t0s + dts * c
    cin = torch.zeros(n_steps, S, 8, dtype=torch.float32)
    for i in range(S):
        for j, a in enumerate(tab.A[i]):
            cin[:, i, j] = dts * a
    cout = torch.zeros(n_steps, S, 8, dtype=torch.float32)
    for j, b in enumerate(tab.b):
        cout[:, S - 1, j] = dts * b
    slot = torch.arange(S, dtype=torch.int32).repeat(n_steps, 1)
    flags = torch.zeros(n_steps, S, dtype=torch.int32)
    flags[:, S - 1] = FLAG_STEP_END
    return EvalPlan(
        t_eval=(sign * tau).reshape(n_evals),
        sign=sign,
        slot=slot.reshape(n_evals),
        flags=flags.reshape(n_evals),
        cin=cin.reshape(n_evals, 8),
        cout=cout.reshape(n_evals, 8),
        n_steps=n_steps,
    )


def build_table(plan: EvalPlan, a: torch.Tensor, b: torch.Tensor, c1: torch.Tensor, width: int,
                gn: Optional[torch.Tensor] = None, noise_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Pack the rows: header words + first-layer bias ``c1`` zero-padded to the on-chip width.

    ``a``/``b`` are the RHS coefficients in *real* time; the reversal sign of the plan is
    applied here.
    """
    n = plan.t_eval.numel()
    rows = torch.zeros(n, ROW_HDR + width, dtype=torch.float32)
    rows[:, 0] = plan.sign * a.to(torch.float32)
    rows[:, 1] = plan.sign * b.to(torch.float32)
    if gn is not None:
        rows[:, 2] = gn
    iview = rows.view(torch.int32)
    iview[:, 3] = plan.flags
    iview[:, 4] = plan.slot
    if noise_idx is not None:
        iview[:, 5] = noise_idx.to(torch.int32)
    rows[:, 8:16] = plan.cin
    rows[:, 16:24] = plan.cout
    rows[:, ROW_HDR:ROW_HDR + c1.shape[1]] = c1.to(torch.float32)
    return rows


def plan_euler_maruyama(T, epsilon, steps: int):
    """Times and step of the reverse-SDE loop (diffusion.py:539-559), in fp32 like the reference.

    Returns (t_values [n_done] fp32, dt 0-dim fp32).  The reference stops early if the
    accumulated time falls below epsilon (diffusion.py:548-551).
    """
    T = torch.as_tensor(T, dtype=torch.float32)
    epsilon = torch.as_tensor(epsilon, dtype=torch.float32)
    dt = -(T - epsilon) / steps
    t = torch.ones(1, dtype=torch.float32) * T
    ts = []
    for _ in range(steps):
        if bool(t[0] < epsilon):
            break
        ts.append(t[0].clone())
        t = t + dt
    return torch.stack(ts) if ts else torch.zeros(0), dt
