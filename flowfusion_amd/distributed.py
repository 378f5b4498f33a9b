"""Multi-GPU execution of the sampling / log-density path: one process per GPU, batch sharding,
one collective at the end (adaptive solves: plus the exchange of their error norms, `global_step_control`).

The reference is single-process (no torch.distributed anywhere in flowfusion/).  Samples are
independent -- nothing on the path couples two rows of the batch -- so the batch axis is cut into
contiguous, balanced shards, every rank integrates its shard with the fused kernel (weights and the
evaluation table are a few MB and simply replicated), and the results meet in a single RCCL
all-gather over xGMI (backend "nccl" is RCCL on ROCm).  Fixed-grid solves use no other collective.  Adaptive solves have
one real exchange step: torchdiffeq chooses ONE step size for the whole batch from a norm over all of it, so the sums of
squares behind each norm are all-reduced (8 doubles) when a batch is cut over ranks -- `global_step_control`.
"""
from __future__ import annotations

import contextlib
import contextvars
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

# ---- the one exchange step of the path: step control of an ADAPTIVE solve over several shards of one batch -----------------
# (a context variable: the setting belongs to the solve that entered the context, not to every host thread of the process)
_STEP_CONTROL: contextvars.ContextVar = contextvars.ContextVar("ff_step_control", default=(False, None))


@contextlib.contextmanager
def global_step_control(group=None):
    """Adaptive solves inside this context control their step size from the error norm of the WHOLE batch, as torchdiffeq
    does for the batch it is handed (one step size for all samples: diffusion.py:631-639, 744-752; flow.py:299-303,
    371-382), although each rank holds only a shard: the sums of squares behind every norm are summed over ``group``
    (an RCCL all-reduce of 8 doubles per norm, enqueued between the reduction kernel and the controller kernel -- no host
    synchronisation on the device-side controller).  Every rank then takes the same accept / reject decisions and the same
    steps, so a sample's result does not depend on the sharding beyond the rounding of those sums.  Every rank must enter
    the same solves in the same order, each with at least one row.  Outside the context (the default) a rank controls its
    steps from its own rows.  Fixed-grid solves need none of this."""
    if not dist.is_initialized():
        raise RuntimeError("global_step_control needs an initialised torch.distributed process group")
    token = _STEP_CONTROL.set((True, group))
    try:
        yield
    finally:
        _STEP_CONTROL.reset(token)


def step_control_group():
    """(active, group) of the enclosing ``global_step_control`` context."""
    return _STEP_CONTROL.get()


def _is_adaptive(method) -> bool:
    from . import solvers
    return method in solvers.ALL_ADAPTIVE


def _step_control(n: int, world: int, group, global_control: bool, method):
    """The context a sharded solve runs in.  Batch-global step control needs every rank in the exchange, and an empty
    shard has no launch to hang the exchange on: its peers would wait in the all-reduce for ever.  ``n`` and ``world``
    are known to every rank, so every rank raises here, before anyone enters a collective."""
    if not (global_control and world > 1 and _is_adaptive(method)):
        return contextlib.nullcontext()
    if n < world:
        raise ValueError(f"global step control over {world} ranks needs at least one row per rank, the batch has {n}: use fewer "
                         "ranks, a fixed-grid method, or global_control=False (every rank then steps from its own rows)")
    return global_step_control(group)


def sum_over_ranks_(values: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum of a small tensor over the ranks, ordered with the current stream.  RCCL reduces device tensors in
    place; the gloo rehearsal (several ranks on one card, CPU tests) goes through the host."""
    if dist.get_backend(group) == "nccl" or not values.is_cuda:
        dist.all_reduce(values, group=group)
    else:
        host = values.cpu()
        dist.all_reduce(host, group=group)
        values.copy_(host)
    return values


def sum_over_ranks(values: List[float], device, group=None) -> List[float]:
    """Host-side form (the host step controller): python floats in, their sums over the ranks out."""
    on = device if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor(values, dtype=torch.float64, device=on)
    dist.all_reduce(t, group=group)
    return [float(v) for v in t.cpu()]


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) of an n-row batch owned by `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n: int, world: int):
    return [shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0] for r in range(world)]


def gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather the row shards produced under `shard_bounds` back into the full [n_total, ...]
    tensor on every rank (one collective; ragged shards are padded to the largest)."""
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_total, world)
    if world == 1:
        return local
    if local.is_cuda and dist.get_backend(group) != "nccl":
        # (the gloo rehearsal of the multi-rank path -- several ranks on one card, CPU tests -- gathers through the host)
        return gather_rows(local.cpu(), n_total, group).to(local.device)
    if len(set(sizes)) == 1:
        out = local.new_empty((n_total,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    mx = max(sizes)
    pad = local.new_zeros((mx,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    buf = local.new_empty((world * mx,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


def run_sharded(fn: Callable[..., torch.Tensor], inputs: Sequence[Optional[torch.Tensor]], group=None,
                gather: bool = True):
    """Apply `fn(*row_slices)` to this rank's shard of every [B, ...] input and (optionally) gather.

    Every rank passes the same full-batch `inputs` (or `None` entries); rank r computes rows
    ``shard_bounds(B, world, r)``.  With ``gather=False`` the local result is returned together
    with its bounds, for consumers that keep the data distributed.
    """
    n = next(t.shape[0] for t in inputs if t is not None)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(n, world, rank)
    local = fn(*[None if t is None else t[lo:hi].contiguous() for t in inputs])
    if not gather:
        return local, (lo, hi)
    if isinstance(local, (tuple, list)):
        return type(local)(gather_rows(t, n, group) if torch.is_tensor(t) else t for t in local)
    return gather_rows(local, n, group)


def sample_sde_sharded(score_model, shape, conditional: Optional[torch.Tensor] = None, steps: int = 100,
                       seed: int = 0, group=None, gather: bool = True, local_conditional: Optional[torch.Tensor] = None):
    """Euler-Maruyama sampling (``ScoreModel.sample_sde``) of a [B, dim] batch over all ranks, with the
    same result for any number of ranks: both the prior draw and the per-step noise come from the library's
    counter-based stream keyed by ``seed`` and the GLOBAL row index (prior: ``ff_normal_fill`` with the reserved
    noise index; steps: ``noise="philox"``), so a rank only ever touches its own rows.  One all-gather at the
    end.  ``conditional`` is the full [B, C] tensor (every rank slices its rows); ``local_conditional`` is this rank's
    [hi - lo, C] rows already (a consumer that never materialises the full batch)."""
    from . import _native
    batch, *dims = shape
    if len(dims) != 1:
        raise NotImplementedError("sample_sde_sharded: only [batch, dim] states are supported")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(batch, world, rank)
    dev = next(score_model.model.parameters()).device
    # the prior is Normal(0, scale) (diffusion.py:1003, 1093)
    sde = score_model.sde
    scale = float(sde.sigma_max) if hasattr(sde, "sigma_max") else 1.0
    x = _native.normal_fill(hi - lo, dims[0], int(seed), lo, dev, scale=scale)
    cond = None if conditional is None else conditional[lo:hi].contiguous()
    if local_conditional is not None:
        if conditional is not None or local_conditional.shape[0] != hi - lo:
            raise ValueError("local_conditional must hold exactly this rank's rows (and excludes `conditional`)")
        cond = local_conditional.contiguous()
    local = score_model._sample_sde_from(x, None, cond, steps, rng=(int(seed), lo))
    if not gather:
        return local, (lo, hi)
    return gather_rows(local, batch, group) if world > 1 else local


def log_prob_sharded(score_model, x: Optional[torch.Tensor] = None, conditional: Optional[torch.Tensor] = None,
                     seed: int = 0, group=None, gather: bool = True, local_x: Optional[torch.Tensor] = None,
                     n_total: Optional[int] = None, local_conditional: Optional[torch.Tensor] = None,
                     global_control: bool = True, **solver):
    """``ScoreModel.log_prob`` of a [B, dim] batch over all ranks; one all-gather of the [B, 1] result at the end.

    ``x`` (and ``conditional``) are the full tensors, every rank slicing its rows -- or ``local_x`` (and
    ``local_conditional``) are this rank's rows already, with ``n_total`` the size of the whole batch.  A Hutchinson,
    Hutch++ or XTrace model takes its probes from the library's counter-based stream keyed by ``seed`` and the GLOBAL row
    (``probe="philox"``), so with a fixed-grid ``method`` a row's result does not depend on the number of ranks; the
    exact trace needs no random numbers.  ``**solver`` (atol, rtol, method, options) goes to ``log_prob`` unchanged.
    Under an adaptive ``method`` (the reference's default) the step size comes from the error norm of the WHOLE batch, as
    torchdiffeq takes it (``global_step_control``: one small all-reduce per norm; every rank needs at least one row);
    results then agree across world sizes to the rounding of those norms.  ``global_control=False``: every rank
    controls its steps from its own rows (agreement to the solver tolerances only)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if (x is None) == (local_x is None):
        raise ValueError("pass either the full batch `x` or this rank's rows `local_x` (with n_total)")
    if x is not None:
        n = x.shape[0]
        lo, hi = shard_bounds(n, world, rank)
        rows = x[lo:hi].contiguous()
        if local_conditional is not None:
            raise ValueError("local_conditional goes with local_x")
        cond = None if conditional is None else conditional[lo:hi].contiguous()
    else:
        if n_total is None or conditional is not None:
            raise ValueError("local_x needs n_total (and local_conditional instead of conditional)")
        n = int(n_total)
        lo, hi = shard_bounds(n, world, rank)
        if local_x.shape[0] != hi - lo or (local_conditional is not None and local_conditional.shape[0] != hi - lo):
            raise ValueError(f"rank {rank} of {world} owns rows [{lo}, {hi}) of {n}: local tensors must hold exactly those")
        rows = local_x.contiguous()
        cond = None if local_conditional is None else local_conditional.contiguous()
    randomised = any(getattr(score_model, k, False) for k in ("hutch", "hutchpp", "xtrace"))
    extra = {"probe": "philox", "seed": int(seed), "sample_offset": lo} if randomised else {}
    with _step_control(n, world, group, global_control, solver.get("method", "dopri5")):
        local = score_model.log_prob(rows, conditional=cond, **solver, **extra)
    if not gather:
        return local, (lo, hi)
    return gather_rows(local, n, group) if world > 1 else local


def sample_ode_sharded(score_model, n_total: int, dim: int, seed: int = 0, conditional: Optional[torch.Tensor] = None,
                       group=None, gather: bool = True, local_conditional: Optional[torch.Tensor] = None,
                       global_control: bool = True, **solver):
    """``ScoreModel.sample_ode_from_base`` of ``n_total`` base samples over all ranks: the base samples are standard
    normals of the library's counter-based stream keyed by ``seed`` and the GLOBAL row (``ff_normal_fill``), so every world
    size transports the same points and a rank only ever touches its rows; one all-gather at the end.  ``**solver`` (atol,
    rtol, method, options) goes to ``sample_ode_from_base`` unchanged; with an adaptive method (the reference's default)
    and ``global_control`` the step size comes from the error norm of the whole batch (``global_step_control``)."""
    from . import _native
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(int(n_total), world, rank)
    dev = next(score_model.model.parameters()).device
    z = _native.normal_fill(hi - lo, int(dim), int(seed), lo, dev)
    cond = None if conditional is None else conditional[lo:hi].contiguous()
    if local_conditional is not None:
        if conditional is not None or local_conditional.shape[0] != hi - lo:
            raise ValueError("local_conditional must hold exactly this rank's rows (and excludes `conditional`)")
        cond = local_conditional.contiguous()
    with _step_control(int(n_total), world, group, global_control, solver.get("method", "dopri5")):
        local, _ = score_model.sample_ode_from_base(z, conditional=cond, **solver)
    if not gather:
        return local, (lo, hi)
    return gather_rows(local, int(n_total), group) if world > 1 else local


# ---- the flows (flowfusion/flow.py:259-306, 386-438; 750-799, 885-941) -- BASELINE configs[3] is worded "sharded over 8xMI355X"
def _local_rows(full, local, lo, hi, what):
    if local is not None:
        if full is not None or local.shape[0] != hi - lo:
            raise ValueError(f"local_{what} must hold exactly this rank's rows [{lo}, {hi}) (and excludes `{what}`)")
        return local.contiguous()
    return None if full is None else full[lo:hi].contiguous()


def flow_sample_sharded(flow, n_total: int, seed: int = 0, conditional: Optional[torch.Tensor] = None, group=None,
                        gather: bool = True, local_conditional: Optional[torch.Tensor] = None, global_control: bool = True,
                        **solver):
    """``ODEFlow.sample`` / ``ConditionalODEFlow.sample`` of ``n_total`` base samples over all ranks.  The base samples
    are standard normals of the library's counter-based stream keyed by ``seed`` and the GLOBAL row (``ff_normal_fill``):
    every world size transports the same points and a rank only ever touches its own rows.  ``conditional`` is the raw
    [n_total, C] tensor (every rank slices its rows) or ``local_conditional`` this rank's rows; the flow normalises it
    itself (flow.py:771-777).  One all-gather at the end (north_star: "a single RCCL gather"; ``gather=False`` keeps the
    shard and returns its bounds).  ``**solver`` (method, options, atol, rtol) goes to ``sample`` unchanged; under an
    adaptive method -- the reference's default -- the step size comes from the error norm of the whole batch
    (``global_step_control``), ``global_control=False``: from the rank's own rows."""
    from . import _native
    from .flow import _DEFAULT_SAMPLE_METHOD
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = int(n_total)
    lo, hi = shard_bounds(n, world, rank)
    dev = next(flow.parameters()).device
    xT = _native.normal_fill(hi - lo, int(flow.target_dimension), int(seed), lo, dev)
    cond = _local_rows(conditional, local_conditional, lo, hi, "conditional")
    method = solver.get("method") or _DEFAULT_SAMPLE_METHOD
    with _step_control(n, world, group, global_control, method):
        local = flow.sample(xT, **solver) if cond is None else flow.sample(xT, cond, **solver)
    if not gather:
        return local, (lo, hi)
    return gather_rows(local, n, group) if world > 1 else local


def flow_log_prob_sharded(flow, x: Optional[torch.Tensor] = None, conditional: Optional[torch.Tensor] = None, seed: int = 0,
                          group=None, gather: bool = True, local_x: Optional[torch.Tensor] = None,
                          n_total: Optional[int] = None, local_conditional: Optional[torch.Tensor] = None,
                          global_control: bool = True, **solver):
    """``ODEFlow.log_prob`` / ``ConditionalODEFlow.log_prob`` of a [B, dim] batch over all ranks; one all-gather of the
    [B] result at the end.  ``x`` (and ``conditional``) are the full tensors, every rank slicing its rows -- or ``local_x``
    (and ``local_conditional``) this rank's rows already, with ``n_total`` the size of the whole batch.  With
    ``hutchinson=True`` the probe comes from the library's counter-based stream keyed by ``seed`` and the GLOBAL row
    (``probe="philox"``), so a row's result does not depend on the number of ranks; the exact trace (the reference's
    default, flow.py:158-161) needs no random numbers.  Step control as in ``flow_sample_sharded``."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if (x is None) == (local_x is None):
        raise ValueError("pass either the full batch `x` or this rank's rows `local_x` (with n_total)")
    if x is None and n_total is None:
        raise ValueError("local_x needs n_total")
    n = int(x.shape[0] if x is not None else n_total)
    lo, hi = shard_bounds(n, world, rank)
    rows = _local_rows(x, local_x, lo, hi, "x")
    if x is not None and local_conditional is not None:
        raise ValueError("local_conditional goes with local_x")
    if x is None and conditional is not None:
        raise ValueError("local_x goes with local_conditional, not the full `conditional`")
    cond = _local_rows(conditional, local_conditional, lo, hi, "conditional")
    extra = {"probe": "philox", "seed": int(seed), "sample_offset": lo} if solver.get("hutchinson") else {}
    with _step_control(n, world, group, global_control, solver.get("method", "dopri5")):
        local = flow.log_prob(rows, **solver, **extra) if cond is None else flow.log_prob(rows, cond, **solver, **extra)
    if not gather:
        return local, (lo, hi)
    return gather_rows(local, n, group) if world > 1 else local
