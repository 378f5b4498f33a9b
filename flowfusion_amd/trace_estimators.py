"""Hutch++ and XTrace divergence estimators on per-sample Jacobians.

The reference evaluates them inside ``ScoreModel.forward`` (diffusion.py:336-481) through reverse-mode
products ``A v = J^T v`` (``torch.func.vjp`` + ``vmap``), with J = d xdot / d x.  On the fused path one
launch in exact mode returns the whole matrix ``A[b] = J[b]^T`` (ff_ode_args.jac_out: row j = J^T e_j),
so every product below is a small batched matmul; the formulas -- which products are taken, the QR
factorisations, the leave-one-out combination of XTrace -- follow the reference line by line in meaning,
so that given the same probes both give the same estimate up to fp32 rounding.

All functions take ``A`` of shape [B, D, D] and probe tensors laid out like the reference's
(``[n_probes, B, D]``, entries +-1), and return the divergence estimate [B].

Since round 4 the fused path evaluates the estimators in its own kernel (csrc/ff_trace_est.h / ff_trace.hip: the same
formulas, one launch for every evaluation row); this module stays as the torch statement of them -- pinned by the
reference's fixtures in the CPU tier, the comparator of the kernel in both tiers, what ``ScoreModel.forward`` uses on CPU
tensors, and (``FF_TORCH_ESTIMATOR=1``) rounds 1-3's route for A/B runs.
"""
from __future__ import annotations

import torch


def _cols(V: torch.Tensor) -> torch.Tensor:
    """Probes [n, B, D] -> matrix with the probes as columns [B, D, n]."""
    return V.permute(1, 2, 0)


def thin_qr(Y: torch.Tensor):
    """Reduced QR of a batch of tall matrices [N, D, r] with a handful of columns, by Householder reflections in
    LAPACK's conventions (``geqrf`` + ``orgqr``: beta = -sign(alpha) |x|, tau = (beta - alpha) / beta, H = I when the
    part of the column below the diagonal vanishes), written as a few batched elementwise operations per column.
    Why not ``torch.linalg.qr``: on ROCm it takes ~27 us PER MATRIX at these sizes -- 23 s for the 10^6 sketches of a
    2,500-sample, 400-evaluation solve.  Why Householder and not Gram-Schmidt: a sketch A S with +-1 probes is rank
    deficient now and then (two probes equal up to sign), and the reference's Q then still has r orthonormal columns
    (the reflectors complete the basis); the estimate depends on that completion.  Falls back to ``torch.linalg.qr``
    beyond 8 columns."""
    N, D, r = Y.shape
    if r > 8 or r > D:
        return torch.linalg.qr(Y, mode="reduced")
    Wk = Y.clone()                                   # becomes R in its upper triangle
    vs, taus = [], []
    for j in range(r):
        x = Wk[:, j:, j]                             # [N, D - j]
        alpha = x[:, 0]
        xnorm = torch.linalg.vector_norm(x[:, 1:], dim=1) if D - j > 1 else torch.zeros_like(alpha)
        nontrivial = xnorm > 0
        beta = -torch.copysign(torch.sqrt(alpha * alpha + xnorm * xnorm), alpha)
        beta = torch.where(nontrivial, beta, alpha)
        safe_beta = torch.where(nontrivial, beta, torch.ones_like(beta))
        tau = torch.where(nontrivial, (beta - alpha) / safe_beta, torch.zeros_like(alpha))
        denom = torch.where(nontrivial, alpha - beta, torch.ones_like(alpha))
        v = torch.cat([torch.ones_like(alpha)[:, None], x[:, 1:] / denom[:, None]], dim=1)      # v[0] = 1
        v = torch.where(nontrivial[:, None], v, torch.zeros_like(v))
        if j + 1 < r:                                # apply H = I - tau v v^T to the remaining columns
            C = Wk[:, j:, j + 1:]
            Wk[:, j:, j + 1:] = C - tau[:, None, None] * v[:, :, None] * (v[:, :, None] * C).sum(dim=1, keepdim=True)
        Wk[:, j, j] = beta
        vs.append(v)
        taus.append(tau)
    R = torch.triu(Wk[:, :r, :])
    Q = torch.eye(D, r, dtype=Y.dtype, device=Y.device).expand(N, D, r).clone()
    for j in range(r - 1, -1, -1):
        v, tau = vs[j], taus[j]
        C = Q[:, j:, :]
        Q[:, j:, :] = C - tau[:, None, None] * v[:, :, None] * (v[:, :, None] * C).sum(dim=1, keepdim=True)
    return Q, R


def hutchpp(A: torch.Tensor, S: torch.Tensor, G: torch.Tensor) -> torch.Tensor:
    """Hutch++ (Meyer, Musco, Musco, Woodruff 2021) as the reference runs it (diffusion.py:336-399):
    sketch Y = A S, orthonormal basis Q of its range, exact trace on that range plus a Hutchinson
    estimate on the deflated remainder with the probes G."""
    Y = A @ _cols(S)                                            # [B, D, r]
    Q, _ = thin_qr(Y)                                           # [B, D, k]
    AQ = A @ Q
    trace_range = (Q * AQ).sum(dim=(1, 2))                      # sum_i q_i^T A q_i
    Gc = _cols(G)                                               # [B, D, m]
    U = Gc - Q @ (Q.transpose(1, 2) @ Gc)                       # (I - Q Q^T) G
    AU = A @ U
    trace_rest = (U * AU).sum(dim=(1, 2)) / float(G.shape[0])
    return trace_range + trace_rest


def xtrace(A: torch.Tensor, O: torch.Tensor) -> torch.Tensor:
    """XTrace (Epperly, Tropp, Webber 2024) as the reference runs it (diffusion.py:401-481): for every
    probe i a Hutch++-like estimate whose basis leaves probe i out, obtained from ONE QR of Y = A O by a
    rank-one downdate (the columns of S below are the normalised rows of R^-1), averaged over i."""
    Oc = _cols(O)                                               # [B, D, m]
    Y = A @ Oc
    Q, R = thin_qr(Y)                                           # [B, D, k], [B, k, m] with k = m
    k = Q.shape[2]
    Z = A @ Q                                                   # [B, D, k]
    H = Q.transpose(1, 2) @ Z                                   # Q^T A Q
    W = Q.transpose(1, 2) @ Oc                                  # Q^T O      [B, k, m]
    T = Z.transpose(1, 2) @ Oc                                  # (A Q)^T O  [B, k, m]
    eye = torch.eye(k, device=A.device, dtype=A.dtype)
    Rinv = torch.linalg.solve_triangular(R, eye, upper=True)    # [B, k, k]
    Rinv = Rinv / torch.linalg.vector_norm(Rinv, dim=-1, keepdim=True)     # rows to unit length
    S = Rinv.transpose(1, 2)                                    # column i = direction removed for probe i
    X = W - (S * W).sum(dim=1, keepdim=True) * S                # w_i - (s_i^T w_i) s_i, column by column
    tr_h = torch.diagonal(H, 0, 1, 2).sum(dim=-1)               # [B]
    shs = (S * (H @ S)).sum(dim=1)                              # s_i^T H s_i   [B, k]
    xhx = (X * (H @ X)).sum(dim=1)                              # x_i^T H x_i
    ws = (W * S).sum(dim=1)
    sr = (S * R).sum(dim=1)
    tx = (T * X).sum(dim=1)
    per_probe = tr_h[:, None] - shs + ws * sr - tx + xhx
    return per_probe.mean(dim=1)


def draw_probes(n: int, like: torch.Tensor) -> torch.Tensor:
    """``torch.sign(torch.randn(n, B, D))`` on the state's device and dtype, as the reference draws S, G and O
    (diffusion.py:708-719)."""
    B, D = like.shape
    return torch.sign(torch.randn(n, B, D, device=like.device, dtype=like.dtype))


def draw_probes_philox(n: int, like: torch.Tensor, seed: int, sample_offset: int, second_set: bool = False) -> torch.Tensor:
    """[n, B, D] +-1 probes from the library's counter-based stream (``ff_normal_fill`` with the reserved indices
    FF_TRACE_PROBE_NOISE_BASE + c): probe c of global row ``sample_offset + r`` is the sign of that row's normal -- drawn on
    the device and independent of how a batch is cut into shards.  ``second_set`` = the residual probes G of Hutch++."""
    from . import _native
    B, D = like.shape
    base = _native.TRACE_PROBE_NOISE_BASE + (0x8000 if second_set else 0)
    if n > 0x8000:
        raise ValueError("at most 32768 probes per set")
    zs = [_native.normal_fill(B, D, seed, sample_offset, like.device, noise_index=base + c) for c in range(n)]
    return torch.where(torch.stack(zs) >= 0, 1.0, -1.0).to(torch.float32)
