"""Hutch++ / XTrace solves outside the device-resident adaptive path (device_adaptive.py, which is where the reference's
default-argument calls run since round 4).

* ``run_table_recorded``: FIXED grids -- the state never depends on the divergence, so the whole table runs as one fused
  launch per tangent pass with the Jacobian of every row recorded (``ff_ode_args.jac_all``), ONE ``ff_trace_estimate``
  launch turns them into estimates, and the rows are combined with the tableau's weights.
* ``rhs_div`` / ``make_step`` / ``run_table``: row-by-row execution -- each right-hand side one launch
  (``flowfusion_amd::mlp_rhs_jac``: the value and its full Jacobian from unit tangents, the network still fused), the
  estimate by ``div_fn``, the Runge-Kutta bookkeeping (the kernel's ``cin`` / ``cout`` / slot semantics) on the host.  This
  is the HOST-controller route of an adaptive estimator solve (``FF_HOST_CONTROLLER=1``, SDE classes the device controller
  does not know, Jacobians beyond the memory budget) and what the CPU tests drive with the kernel-semantics emulator.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import _native
from ._native import MODE_EXACT
from .fused import FusedNet, exact_trace_passes
from .solvers import FLAG_STEP_END, MAX_SLOTS, ROW_HDR


class RowStepper:
    def __init__(self, net: FusedNet, device, cond: Optional[torch.Tensor],
                 div_fn: Callable[[torch.Tensor], torch.Tensor], launcher=None):
        """``div_fn(A)`` maps A[b] = J[b]^T ([B, D, D]) to the divergence estimate [B].  ``launcher(y, rows, first,
        count, jac) -> rhs`` replaces the GPU launch in the CPU tests."""
        self.net = net
        self.device = device
        self.div_fn = div_fn
        plan = net.plan(MODE_EXACT)
        self.width = int(plan.width)
        self.passes = exact_trace_passes(net.dim, plan.tile)
        f32 = lambda t: None if t is None else t.detach().to(device, torch.float32).contiguous()
        cond_d = f32(cond) if net.cond_dim > 0 else None
        if launcher is None:
            words = _native.plan_words(plan)
            wpack = net.wpack(device, MODE_EXACT)
            launcher = lambda y, rows, first, count, jac: torch.ops.flowfusion_amd.mlp_rhs_jac(
                y, cond_d, wpack, rows.to(device), words, first, count, jac)
        self.launcher = launcher
        self.n_evals = 0

    def rhs_div(self, y: torch.Tensor, a: float, b: float, c1: torch.Tensor):
        """(rhs [B, D], divergence estimate [B]) of  a*y + b*NET(y; c1)."""
        rows = torch.zeros(3, ROW_HDR + self.width, dtype=torch.float32)
        rows[0, 0], rows[0, 1] = a, b
        rows[0, ROW_HDR:ROW_HDR + c1.numel()] = c1
        rows[1, 8] = 1.0                               # auxiliary output 0 = stage slot 0 = the right-hand side
        y = y.contiguous()
        jac = torch.empty(y.shape[0], self.net.dim, self.net.dim, dtype=torch.float32, device=y.device)
        rhs = None
        for first, count in self.passes:
            rhs = self.launcher(y, rows, first, count, jac)
        self.n_evals += 1
        return rhs, self.div_fn(jac)

    def run_table(self, x: torch.Tensor, table: torch.Tensor):
        """Fixed-grid integration over the rows of a (CPU) evaluation table; returns (x_final, dlogp [B])."""
        ints = table.view(torch.int32)
        ks = [torch.zeros_like(x) for _ in range(MAX_SLOTS)]
        kl = [torch.zeros(x.shape[0], device=x.device) for _ in range(MAX_SLOTS)]
        lp = torch.zeros(x.shape[0], device=x.device)
        for e in range(table.shape[0]):
            cin, cout = table[e, 8:8 + MAX_SLOTS], table[e, 16:16 + MAX_SLOTS]
            y = x.clone()
            for s in range(MAX_SLOTS):
                if float(cin[s]) != 0.0:
                    y = y + float(cin[s]) * ks[s]
            slot = int(ints[e, 4])
            ks[slot], kl[slot] = self.rhs_div(y, float(table[e, 0]), float(table[e, 1]), table[e, ROW_HDR:])
            if int(ints[e, 3]) & FLAG_STEP_END:
                for s in range(MAX_SLOTS):
                    if float(cout[s]) != 0.0:
                        x = x + float(cout[s]) * ks[s]
                        lp = lp + float(cout[s]) * kl[s]
        return x, lp

    def run_table_recorded(self, x: torch.Tensor, table: torch.Tensor, div_rows, cond=None, max_bytes: int = 1 << 30):
        """Fixed-grid integration WITHOUT host stepping: the state never depends on the divergence, so the whole table
        runs as one fused launch per tangent pass with the Jacobian of every row recorded (ff_ode_args.jac_all), the
        estimator is evaluated for all rows at once (``div_rows(A [n, b, D, D], lo, hi) -> [n, b]`` for samples
        lo..hi), and the estimates are combined with the weights the tableau gives each row.  The batch is cut so that
        a chunk's Jacobians stay under ``max_bytes``.  Returns (x_final, dlogp [B])."""
        n, D, B = int(table.shape[0]), self.net.dim, int(x.shape[0])
        ints = table.view(torch.int32)
        # weight of row e's divergence in the integral: the step-update coefficients applied to the slot it sits in
        w = torch.zeros(n, dtype=torch.float64)
        owner = {}
        for e in range(n):
            owner[int(ints[e, 4])] = e
            if int(ints[e, 3]) & FLAG_STEP_END:
                for s in range(MAX_SLOTS):
                    c = float(table[e, 16 + s])
                    if c != 0.0 and s in owner:
                        w[owner[s]] += c
        w = w.to(torch.float32).to(x.device)
        dev = x.device
        plan = self.net.plan(MODE_EXACT)
        words = _native.plan_words(plan)
        wpack = self.net.wpack(dev, MODE_EXACT)
        tab = table.to(dev)
        f32 = lambda t: None if t is None else t.detach().to(dev, torch.float32).contiguous()
        cond_d = f32(cond) if self.net.cond_dim > 0 else None
        chunk = max(1, min(B, max_bytes // max(1, n * D * D * 4)))
        outs, lps = [], []
        for lo in range(0, B, chunk):
            hi = min(B, lo + chunk)
            xc = x[lo:hi].contiguous()
            cc = None if cond_d is None else cond_d[lo:hi].contiguous()
            jac = torch.empty(n, hi - lo, D, D, dtype=torch.float32, device=dev)
            for first, count in self.passes:
                xo = torch.ops.flowfusion_amd.mlp_ode_jacobians(xc, cc, wpack, tab, words, first, count, jac)
            outs.append(xo)
            lps.append((w[:, None] * div_rows(jac, lo, hi)).sum(dim=0))
            self.n_evals += n
        return torch.cat(outs), torch.cat(lps)

    def make_step(self, schedule, sign: float):
        """Step function for ``adaptive.Dopri5`` (same contract as ``FusedNet.make_step``)."""
        def step(y, k1, lp0, kl1, t_rows, cin, slots, tail, use_y, n_aux):
            a, b, c1 = schedule(sign * t_rows)
            ks = [torch.zeros_like(y) for _ in range(MAX_SLOTS)]
            kl = [torch.zeros(y.shape[0], device=y.device) for _ in range(MAX_SLOTS)]
            if k1 is not None:
                ks[0] = k1
            if kl1 is not None:
                kl[0] = kl1
            for i in range(int(t_rows.numel())):
                yi = y.clone()
                for s in range(MAX_SLOTS):
                    if float(cin[i, s]) != 0.0:
                        yi = yi + float(cin[i, s]) * ks[s]
                slot = int(slots[i])
                ks[slot], kl[slot] = self.rhs_div(yi, float(sign * a[i]), float(sign * b[i]), c1[i])
            aux = torch.zeros(n_aux, *y.shape, device=y.device)
            aux_lp = torch.zeros(n_aux, y.shape[0], device=y.device)
            for j in range(n_aux):
                if (use_y >> j) & 1:
                    aux[j] = aux[j] + y
                    if lp0 is not None:
                        aux_lp[j] = aux_lp[j] + lp0
                for s in range(MAX_SLOTS):
                    cj = float(tail[j, s])
                    if cj != 0.0:
                        aux[j] = aux[j] + cj * ks[s]
                        aux_lp[j] = aux_lp[j] + cj * kl[s]
            return aux, aux_lp
        return step
