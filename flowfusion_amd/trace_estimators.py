"""Hutch++ and XTrace divergence estimators on per-sample Jacobians.

The reference evaluates them inside ``ScoreModel.forward`` (diffusion.py:336-481) through reverse-mode
products ``A v = J^T v`` (``torch.func.vjp`` + ``vmap``), with J = d xdot / d x.  On the fused path one
launch in exact mode returns the whole matrix ``A[b] = J[b]^T`` (ff_ode_args.jac_out: row j = J^T e_j),
so every product below is a small batched matmul; the formulas -- which products are taken, the QR
factorisations, the leave-one-out combination of XTrace -- follow the reference line by line in meaning,
so that given the same probes both give the same estimate up to fp32 rounding.

All functions take ``A`` of shape [B, D, D] and probe tensors laid out like the reference's
(``[n_probes, B, D]``, entries +-1), and return the divergence estimate [B].
"""
from __future__ import annotations

import torch


def _cols(V: torch.Tensor) -> torch.Tensor:
    """Probes [n, B, D] -> matrix with the probes as columns [B, D, n]."""
    return V.permute(1, 2, 0)


def hutchpp(A: torch.Tensor, S: torch.Tensor, G: torch.Tensor) -> torch.Tensor:
    """Hutch++ (Meyer, Musco, Musco, Woodruff 2021) as the reference runs it (diffusion.py:336-399):
    sketch Y = A S, orthonormal basis Q of its range, exact trace on that range plus a Hutchinson
    estimate on the deflated remainder with the probes G."""
    Y = A @ _cols(S)                                            # [B, D, r]
    Q, _ = torch.linalg.qr(Y, mode="reduced")                   # [B, D, k]
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
    Q, R = torch.linalg.qr(Y, mode="reduced")                   # [B, D, k], [B, k, m] with k = m
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
