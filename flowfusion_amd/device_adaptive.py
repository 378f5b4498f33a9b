"""Host side of the device-resident adaptive solver (csrc/ff_adaptive.hip, ``ff_mlp_ode_adaptive``).

Every ODE solve of the reference defaults to torchdiffeq's adaptive ``dopri5`` (flowfusion/diffusion.py:572, 649, 762;
flowfusion/flow.py:299-303, 313).  ``adaptive.Dopri5`` keeps torchdiffeq's batch-global step control on the host -- one
launch, one reduction, one read-back and ~0.25 ms of Python (schedule, table, upload) per attempted step.  This module
moves the loop onto the device: the controller kernel decides accept / reject, picks the next step and writes the next
attempt's evaluation rows itself (SDE schedule scalars, time embedding through the first layer), so a whole chunk of
attempted steps is enqueued with ONE C call and the host reads 128 bytes of state back per chunk.  Same control law,
same arithmetic split (float64 time, fp32 state), same attempt / accept counts as the host controller
(``tests/test_gpu_device_adaptive.py`` runs both).

Several shards of one batch (one process per GPU): inside ``distributed.global_step_control`` every norm launch is split in
two and the C driver calls back between them, where the all-reduce of the sums of squares is enqueued
(``ff_adapt_buffers.exchange``); every rank then walks the same steps, still without a host synchronisation per step.

The host controller stays for what the device one cannot describe: right-hand sides evaluated outside the library
(``generic.py``, ``host_stepper.py``), SDE classes other than the reference's three, more than 32 embedding frequencies,
and the CPU kernel-semantics emulator of the tests.  ``FF_HOST_CONTROLLER=1`` forces it (A/B runs).
"""
from __future__ import annotations

import ctypes
import math
import os
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import torch

from . import _native, adaptive
from ._native import MODE_EXACT, MODE_STATE

FIRST_CHUNK = 16          # attempted steps enqueued before the first look at the state
TRACE = None              # diagnostics: a list collects (attempts, accepted, t, dt, last error ratio) after every attempted step
                          # (one attempt per chunk while it is set; scratch/diag_adaptive_pair.py)
TRACE_ROWS = False        # ... and a host copy of the evaluation table the controller wrote for the NEXT attempt (tests)
MAX_CHUNK = 64
POISON = None             # tests: a value the work buffers are filled with before the solve (uninitialised reads show)
MAX_TIME_COLS = 64        # kMaxTimeCols of ff_adaptive.hip


@dataclass
class ScheduleSpec:
    """The time-dependent part of the right-hand side as ``ff_adapt_config`` wants it."""
    sched: int                                  # _native.SCHED_*
    p: Tuple[float, float, float]               # schedule parameters
    no_sigma: bool
    emb_w: Optional[torch.Tensor]               # device [n_emb] fp32 (score networks) or None (flows)
    pi: float
    w0t: torch.Tensor                           # device [h_real, n_tcols] fp32: the first layer's time columns
    b0: torch.Tensor                            # device [h_real] fp32


def host_controller_forced() -> bool:
    return os.environ.get("FF_HOST_CONTROLLER", "") not in ("", "0")


HOST_ONLY_OPTIONS = ("step_t", "jump_t", "norm")      # torchdiffeq options only the host controller serves (adaptive.py)


def supported(spec: Optional[ScheduleSpec], x: torch.Tensor, net=None, mode: int = MODE_STATE, options: Optional[dict] = None) -> bool:
    """The device controller can run this solve: a schedule it knows, at most 64 time columns in the first layer, none of
    the options that need the host (``step_t`` / ``jump_t``: steps ending on given times; ``norm``: a Python callable) and
    -- for the exact trace -- at most FF_ADAPT_MAX_PASSES unit-tangent passes per attempted step (more than 120 dimensions
    on the 16-column tile).  Otherwise the host controller takes it."""
    if spec is None or not x.is_cuda or host_controller_forced() or spec.w0t.shape[1] > MAX_TIME_COLS:
        return False
    if any((options or {}).get(k) is not None for k in HOST_ONLY_OPTIONS):
        return False
    if net is not None and mode == MODE_EXACT and len(list(_passes(net, net.plan(mode)))) > _native.ADAPT_MAX_PASSES:
        return False
    return True


def build_config(spec: ScheduleSpec, sign: float, method: str, rtol: float, atol: float, options: Optional[dict],
                 keep: list) -> _native.AdaptConfig:
    """``ff_adapt_config`` for an embedded pair of ``adaptive.TABLEAUX``; option handling as ``adaptive.Dopri5``.
    ``keep`` collects the tensors whose device pointers the struct carries."""
    if method not in adaptive.TABLEAUX:
        raise NotImplementedError(f"adaptive method {method!r}: the device controller runs {sorted(adaptive.TABLEAUX)}; dopri8 (13 "
                                  "stages, 7 slots on chip) is stepped stage by stage by adaptive.HostSteppedPair")
    tab = adaptive.TABLEAUX[method]
    opts = dict(options or {})
    c = _native.AdaptConfig()
    c.n_stages, c.order = tab.stages, tab.order
    for i, a in enumerate(tab.alpha):
        c.alpha[i] = a
    for i, beta in enumerate(tab.beta):
        for j, v in enumerate(beta):
            c.beta[i][j] = v
    for j in range(tab.stages):
        c.c_sol[j], c.c_mid[j], c.c_err[j] = tab.c_sol[j], tab.c_mid[j], tab.c_error[j]
    c.rtol, c.atol = float(rtol), float(atol)
    c.min_step = float(opts.pop("min_step", 0.0))
    c.max_step = float(opts.pop("max_step", float("inf")))
    first = opts.pop("first_step", None)
    c.first_step = float("nan") if first is None else float(first)
    c.max_num_steps = int(min(opts.pop("max_num_steps", 2 ** 31 - 1), 2 ** 31 - 1))
    if opts.get("dtype") not in (None, torch.float64):          # torchdiffeq's time dtype: float64 is its default and what runs here
        raise NotImplementedError("adaptive option dtype: time is kept in float64 (torchdiffeq's default); other dtypes are not built")
    for k in HOST_ONLY_OPTIONS:
        if opts.get(k) is not None:
            raise NotImplementedError(f"adaptive option {k!r} runs on the host controller (adaptive.py): device_adaptive.supported() says so")
    c.sched, c.no_sigma, c.sign = spec.sched, int(bool(spec.no_sigma)), float(sign)
    for i, v in enumerate(spec.p):
        c.p[i] = float(v)
    if spec.emb_w is not None:
        c.emb_w, c.n_emb = spec.emb_w.data_ptr(), int(spec.emb_w.numel())
        keep.append(spec.emb_w)
    c.pi = float(spec.pi)
    c.w0t, c.b0 = spec.w0t.data_ptr(), spec.b0.data_ptr()
    c.h_real, c.n_tcols = int(spec.w0t.shape[0]), int(spec.w0t.shape[1])
    keep.extend([spec.w0t, spec.b0])
    return c


def _carve(total: torch.Tensor, sizes: Sequence[int]):
    """Views of `sizes` floats each into one flat fp32 allocation, every view 16-byte aligned."""
    out, off = [], 0
    for n in sizes:
        out.append(total[off:off + n])
        off += (n + 3) // 4 * 4
    return out


def estimator_bytes(net, method: str, B: int, kind: str, probes) -> int:
    """Device memory an estimator solve adds to the exact-trace one: the Jacobians of every evaluation row of an attempt,
    their estimates and the factorisation workspace."""
    code, p0, _, r, _ = _native.trace_kind_and_probes(kind, probes)
    rows = adaptive.TABLEAUX[method].stages - 1
    D = net.dim
    return 4 * (rows * B * D * D + rows * B + int(_native.lib().ff_trace_workspace_floats(code, D, r, rows * B)))


def solve(net, spec: ScheduleSpec, sign: float, mode: int, x: torch.Tensor, t0: float, t_end: float, rtol: float,
          atol: float, options: Optional[dict], method: str, cond=None, probe=None, norm_only=(), estimator=None):
    """``odeint(func, state, [t0, t_end], method=, rtol=, atol=, options=)`` with the loop on the device.
    ``estimator = (kind, probes)`` (``"hutchpp"``: probes (S, G); ``"xtrace"``: (O,); MODE_EXACT only): the divergence that
    is integrated is the Hutch++ / XTrace estimate of the reference (diffusion.py:336-481) instead of the exact trace --
    every attempt records the Jacobians of its evaluation rows, one launch estimates them all, one combines them.
    Returns (y [B, D], lp [B] or None, {"attempts", "accepted", "chunks"})."""
    if not x.is_cuda:
        raise RuntimeError("flowfusion_amd integrates on the GPU only: move the model and its inputs to 'cuda' "
                           f"(got a tensor on {x.device}); there is no CPU fallback")
    dev = x.device
    B, D = x.shape
    has_lp = mode != MODE_STATE
    plan = net.plan(mode)
    keep: list = []
    cfg = build_config(spec, sign, method, rtol, atol, options, keep)
    net.require_slots(cfg.n_stages, mode, "this adaptive method")
    from . import distributed
    exchange, group = distributed.step_control_group()
    if B == 0:
        if exchange:
            raise ValueError("global_step_control: every rank needs at least one row of the batch (an empty shard cannot "
                             "take part in the exchange of the error norms)")
        return x.new_empty(0, D), (x.new_empty(0) if has_lp else None), {"attempts": 0, "accepted": 0, "chunks": 0}
    passes = [(0, 0)] if mode != MODE_EXACT else list(_passes(net, plan))
    P = len(passes)
    if P > _native.ADAPT_MAX_PASSES:
        raise NotImplementedError(f"{P} unit-tangent passes per attempted step exceed FF_ADAPT_MAX_PASSES")
    f32 = lambda t: None if t is None else t.detach().to(dev, torch.float32).contiguous()
    cond_d = f32(cond) if net.cond_dim > 0 else None
    probe_d = f32(probe)
    wpack = net.wpack(dev, mode)
    width = plan.width
    nBD, nB = B * D, B
    # one work allocation: state words (32), y, f0, aux[4], scratch_x, etab, then the divergence arrays; the results
    # are tensors of their own (the caller keeps them, the work buffers go back to the allocator)
    sizes = [32, nBD, nBD, nBD, nBD, nBD, nBD, nBD, 8 * (32 + width)]
    if has_lp:
        sizes += [nB] * (2 + 4 + 1) + ([P * 4 * nB] if P > 1 else [])
    flat = torch.empty(sum((n + 3) // 4 * 4 for n in sizes), dtype=torch.float32, device=dev)
    if POISON is not None:
        flat.fill_(POISON)          # tests: nothing may be read from the work buffers before the solve wrote it
    v = _carve(flat, sizes)
    state, y, f0, aux, scratch_x, etab = v[0], v[1], v[2], v[3:7], v[7], v[8]
    out_y = torch.empty(B, D, dtype=torch.float32, device=dev)
    y.copy_(x.detach().to(torch.float32).reshape(-1))
    b = _native.AdaptBuffers()
    b.y, b.f0, b.scratch_x, b.out_y, b.etab, b.state = (t.data_ptr() for t in (y, f0, scratch_x, out_y, etab, state))
    for j in range(4):
        b.aux[j] = aux[j].data_ptr()
    out_lp = None
    if has_lp:
        lp, fl0, aux_lp, scratch_lp = v[9], v[10], v[11:15], v[15]
        out_lp = torch.empty(B, dtype=torch.float32, device=dev)
        lp.zero_()
        b.lp, b.fl0, b.scratch_lp, b.out_lp = lp.data_ptr(), fl0.data_ptr(), scratch_lp.data_ptr(), out_lp.data_ptr()
        for j in range(4):
            b.aux_lp[j] = aux_lp[j].data_ptr()
        if P > 1:
            b.aux_lp_pass = v[16].data_ptr()
    extras = [f32(c).reshape(-1) for c in norm_only if c is not None and c.numel() > 0]
    if len(extras) > 2:
        raise NotImplementedError("more than two norm-only state components")
    for j, c in enumerate(extras):
        b.norm_only[j], b.norm_only_n[j] = c.data_ptr(), c.numel()
    if estimator is not None:
        if mode != MODE_EXACT:
            raise ValueError("a trace estimator replaces the exact trace: mode must be MODE_EXACT")
        kind, probes = estimator
        code, p0, p1, r, m = _native.trace_kind_and_probes(kind, probes)
        if tuple(p0.shape) != (r, B, D) or (p1 is not None and tuple(p1.shape) != (m, B, D)) or r > D:
            raise RuntimeError(f"{kind} probes of shape {tuple(p0.shape)} do not fit a [{B}, {D}] state")
        p0, p1 = f32(p0), f32(p1)
        rows = cfg.n_stages - 1
        est_jac = torch.empty(rows * B * D * D, dtype=torch.float32, device=dev)
        est_div = torch.empty(rows * B, dtype=torch.float32, device=dev)
        est_ws = torch.empty(max(1, int(_native.lib().ff_trace_workspace_floats(code, D, r, rows * B))), dtype=torch.float32, device=dev)
        if POISON is not None:
            est_jac.fill_(POISON), est_div.fill_(POISON), est_ws.fill_(POISON)
        keep.extend([p0, p1, est_jac, est_div, est_ws])
        b.est_kind, b.est_r, b.est_m = code, r, m
        b.est_probes0, b.est_probes1 = p0.data_ptr(), (0 if p1 is None else p1.data_ptr())
        b.est_jac, b.est_div, b.est_workspace = est_jac.data_ptr(), est_div.data_ptr(), est_ws.data_ptr()
    b.n_passes = P
    for j, (first, count) in enumerate(passes):
        b.pass_first[j], b.pass_count[j] = first, count
    base = _native.OdeArgs()
    base.cond = 0 if cond_d is None else cond_d.data_ptr()
    base.probe = 0 if probe_d is None else probe_d.data_ptr()
    base.wpack = wpack.data_ptr()
    base.batch, base.mode = B, mode
    if cond_d is not None and tuple(cond_d.shape) != (B, plan.cond_dim):
        raise RuntimeError(f"cond has shape {tuple(cond_d.shape)}, expected {(B, plan.cond_dim)}")
    if probe_d is not None and tuple(probe_d.shape) != (B, D):
        raise RuntimeError(f"probe has shape {tuple(probe_d.shape)}, expected {(B, D)}")
    hook_error: list = []
    if exchange:
        # the sums of squares behind every norm meet the other ranks' between the reduction and the controller launch:
        # the C driver calls back here, the all-reduce is enqueued behind the reduction on the same stream
        sums = torch.zeros(_native.EXCHANGE_DOUBLES, dtype=torch.float64, device=dev)

        def _hook(_user, _stream):
            try:
                # the all-reduce is ordered with the launches around it only if it goes onto the stream the C driver
                # enqueues on: that is torch's current stream (passed below), and it must still be when the driver calls back
                if (_stream or 0) != torch.cuda.current_stream(dev).cuda_stream:
                    raise RuntimeError("the exchange hook runs on a different stream than the solve's launches")
                distributed.sum_over_ranks_(sums, group)
                return 0
            except Exception as exc:          # noqa: BLE001 -- reported by the caller below, not through the C frame
                hook_error.append(exc)
                return 1
        hook = _native.EXCHANGE_FN(_hook)
        keep.extend([sums, hook])
        b.exchange_sums = sums.data_ptr()
        b.exchange = ctypes.cast(hook, ctypes.c_void_p).value
    L = _native.lib()
    first_chunk = 1 if TRACE is not None else int(os.environ.get("FF_ADAPT_CHUNK", FIRST_CHUNK))
    what, n, chunks = _native.ADAPT_START | _native.ADAPT_FINISH, first_chunk, 0
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        ws, _ = _native.norm_workspace(dev, stream)
        b.norm_workspace = ws.data_ptr()
        while True:
            rc = L.ff_mlp_ode_adaptive(ctypes.byref(plan), ctypes.byref(base), ctypes.byref(cfg), ctypes.byref(b),
                                       float(t0), float(t_end), what, n, ctypes.c_void_p(stream))
            if rc != _native.FF_OK:
                ws[:4].zero_()
                if hook_error:
                    raise RuntimeError("global_step_control: the all-reduce of the error norms failed") from hook_error[0]
                raise _native._err(rc, "ff_mlp_ode_adaptive")
            chunks += 1
            st = _native.AdaptState.from_buffer_copy(state.view(torch.int32).cpu().numpy().tobytes())
            if st.error:
                # torchdiffeq's assertions, worded as the host controller words them (adaptive.Dopri5)
                msg = {_native.ADAPT_ERR_UNDERFLOW: f"underflow in dt {st.dt}",
                       _native.ADAPT_ERR_NONFINITE: "non-finite values in state `y`",
                       _native.ADAPT_ERR_MAXSTEPS: f"max_num_steps exceeded ({st.n_steps}>={cfg.max_num_steps})"}[st.error]
                err = RuntimeError(msg)
                err.solver_stats = {"attempts": int(st.n_attempts), "accepted": int(st.n_accepted), "chunks": chunks}
                raise err
            if TRACE is not None:
                TRACE.append((int(st.n_attempts), int(st.n_accepted), float(st.t), float(st.dt), float(st.last_ratio))
                             + ((etab.view(8, 32 + width).cpu().clone(),) if TRACE_ROWS else ()))
            if st.done:
                break
            if not st.active:
                raise RuntimeError("adaptive controller stopped without finishing")       # cannot happen
            left = (st.t_end - st.t) / st.dt if st.dt > 0 else MAX_CHUNK
            n = 1 if TRACE is not None else int(min(MAX_CHUNK, max(4, math.ceil(left * 1.3) + 2)))
            what = _native.ADAPT_FINISH
    del keep
    stats = {"attempts": int(st.n_attempts), "accepted": int(st.n_accepted), "chunks": chunks}
    return out_y, (out_lp if has_lp else None), stats


def _passes(net, plan):
    from .fused import exact_trace_passes
    return exact_trace_passes(net.dim, plan.tile)
