"""GPU tier: ScoreModel with a score module that is NOT the reference's MLP (the reference accepts any
``nn.Module(t, x, conditional=None)``, diffusion.py:201, 233-238).  The fused kernels cannot hold such a model; the
library still does the stepping (generic.py: the fused path's evaluation plans and adaptive driver, the module
evaluated on the GPU, every Runge-Kutta combination one ff_stage_combine launch).

Known answers that depend on no stepper but the one under test: data ~ N(mu, s^2 I) diffused by
the SDE stays Gaussian, so the score, the log-density along the probability-flow ODE and the transport map are closed
forms.  Plus: a module that merely wraps an MLP must reproduce the fused kernels' results."""
import math

import pytest
import torch
from torch import nn

from flowfusion_amd import _native
from flowfusion_amd import diffusion as D
from tests.test_gpu_parity import DEV, LOGP_TOL, STATE_TOL, _logp_err, _seeded_score_model, _state_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(built_library):
    assert torch.cuda.is_available(), "the gpu tier needs a GPU"


class GaussianScore(nn.Module):
    """sigma-free score of N(mu, s^2 I) diffused to time t: -(x - m_t mu) / ((m_t s)^2 + sigma_t^2)."""

    def __init__(self, sde, mu, s):
        super().__init__()
        self.sde = [sde]                       # not a submodule: the ScoreModel owns the SDE
        self.register_buffer("mu", mu)
        self.s = s

    def moments(self, t):
        """(mean factor, variance) of the marginal.  Written without the square root of marginal_prob_scalars so that it
        stays smooth a little outside [epsilon, 1]: the adaptive solver's last step overshoots the end of the span (and
        interpolates back), like torchdiffeq's."""
        sde = self.sde[0]
        if isinstance(sde, D.VESDE):
            return torch.ones_like(t), self.s ** 2 + sde.sigma(t) ** 2
        lc = sde._log_coeff(t)
        m, q = torch.exp(-0.5 * lc), 1.0 - torch.exp(-lc)
        return m, (m * self.s) ** 2 + (q ** 2 if isinstance(sde, D.SUBVPSDE) else q)

    def forward(self, t, x, conditional=None):
        if t.dim() == 0:
            t = t * torch.ones(x.shape[0], device=x.device)
        m, var = self.moments(t)
        return -(x - m[:, None] * self.mu) / var[:, None]


def _gauss_logpdf(x, mean, var):
    return -0.5 * ((x - mean) ** 2).sum(1) / var - 0.5 * x.shape[1] * torch.log(2 * math.pi * var)


@pytest.mark.parametrize("sde_name,kw", [("VPSDE", {}), ("VESDE", {"epsilon": 1e-3}), ("SUBVPSDE", {})])
def test_gaussian_density_and_transport_through_the_gpu_path(sde_name, kw):
    sde = getattr(D, sde_name)(**kw)
    mu = torch.tensor([0.5, -1.0, 2.0])
    s = 0.6
    sm = D.ScoreModel(GaussianScore(sde, mu, s), sde, no_sigma=True).eval().to(DEV)
    net = sm.model
    torch.manual_seed(0)
    x0 = (mu + s * torch.randn(64, 3)).to(DEV)
    eps = float(sde.epsilon)
    one, e_ = torch.ones(1, device=DEV), torch.full((1,), eps, device=DEV)
    errs = []
    for n in (20, 40):
        opts = {"step_size": (1.0 - eps) / n}
        xT, dlp = sm.solve_odes_forward(x0, method="rk4", options=opts)                 # exact trace by autograd
        assert xT.is_cuda and dlp.shape == (64, 1) and sm.last_solver_stats["evaluations"] == 4 * n
        m1, v1 = net.moments(one)
        me, ve = net.moments(e_)
        lhs = _gauss_logpdf(xT, m1 * net.mu, v1) + dlp.squeeze(1)                        # log p_1(x(1)) + int div
        rhs = _gauss_logpdf(x0, me * net.mu, ve)                                         # log p_eps(x(eps))
        errs.append((lhs - rhs).abs().max().item())
    assert errs[1] < 5e-3 and errs[1] < errs[0] / 4.0, errs                              # converging at high order (fp32 floor)
    # transport map of the probability-flow ODE: affine, closed form; fixed grid and the adaptive default
    z = torch.randn(200, 3, device=DEV)
    base = z if sde_name != "VESDE" else z                                               # sample_ode_from_base scales by sigma_max itself
    m1, v1 = net.moments(one)
    me, ve = net.moments(e_)
    zz = base * sde.sigma_max.to(DEV) if sde_name == "VESDE" else base
    expect = me * net.mu + torch.sqrt(ve / v1) * (zz - m1 * net.mu)
    x, empty = sm.sample_ode_from_base(base, method="rk4", options={"step_size": (1.0 - eps) / 200})
    assert empty == [] and (x - expect).abs().max().item() < 2e-3 * max(1.0, float(expect.abs().max()))
    if sde_name != "VESDE":
        # (the adaptive solver's last step overshoots the end of the span and interpolates back, like torchdiffeq's; VP's
        # sqrt(beta(t)) is NaN below t = -0.005, so the reference's own default only works on VE-type schedules)
        return
    x, _ = sm.sample_ode_from_base(base, atol=1e-6, rtol=1e-6)                           # dopri5 (reference default method)
    assert sm.last_solver_stats["accepted"] >= 3
    assert (x - expect).abs().max().item() < 2e-3 * max(1.0, float(expect.abs().max()))
    # dopri8 around the module (stage-by-stage driver): a closed-form transport map does not care who restated the tableau
    x8, _ = sm.sample_ode_from_base(base, atol=1e-6, rtol=1e-6, method="dopri8")
    assert sm.last_solver_stats["accepted"] >= 2
    assert (x8 - expect).abs().max().item() < 2e-3 * max(1.0, float(expect.abs().max()))
    # Hutchinson and the default log_prob run too (adaptive, exact trace): same density within solver tolerance
    lp = sm.log_prob(x0[:32], atol=1e-6, rtol=1e-6)
    sm.hutch = True
    torch.manual_seed(1)
    lph = sm.log_prob(x0[:32], atol=1e-6, rtol=1e-6)
    assert lp.shape == (32, 1) and torch.isfinite(lp).all()
    # the Gaussian's Jacobian is a multiple of the identity, so e^T J e = tr J for every +-1 probe: the two must agree
    assert (lp - lph).abs().max().item() < 1e-3


class WrappedMLP(nn.Module):
    """An MLP behind a module that does not look like one (no NN / W / pi attributes): takes the generic path."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, t, x, conditional=None):
        return self.inner(t, x, conditional=conditional)


@pytest.mark.parametrize("D_,C,units,sde_name,no_sigma", [(16, 0, [256] * 4, "VPSDE", True), (5, 3, [64, 100], "VESDE", False)])
def test_wrapped_mlp_matches_the_fused_kernels(D_, C, units, sde_name, no_sigma):
    sm, so32, so64 = _seeded_score_model(D_, C, units, sde_name, no_sigma, 23)
    gm = D.ScoreModel(WrappedMLP(sm.model), sm.sde, no_sigma=no_sigma).eval()
    assert not gm._fusable() and sm._fusable()
    B = 150
    torch.manual_seed(2)
    base = torch.randn(B, D_, device=DEV)
    cond = torch.randn(B, C, device=DEV) if C else None
    opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / 25}
    a, _ = sm.sample_ode_from_base(base, conditional=cond, method="rk4", options=opts)
    b, _ = gm.sample_ode_from_base(base, conditional=cond, method="rk4", options=opts)
    assert _state_err(b, a.cpu()) < STATE_TOL
    ref = so64.sample_ode_from_base(base.cpu().double(), None if cond is None else cond.cpu().double(), "rk4", opts).float()
    assert _state_err(b, ref) < STATE_TOL
    # log-density: Hutchinson with the same CPU-drawn probe, and the exact trace
    x0 = base[:40] * 0.5
    c40 = None if cond is None else cond[:40]
    for hutch in (True, False):
        sm.hutch = gm.hutch = hutch
        torch.manual_seed(9)
        la = sm.log_prob(x0, conditional=c40, method="midpoint", options=opts)
        torch.manual_seed(9)
        lb = gm.log_prob(x0, conditional=c40, method="midpoint", options=opts)
        assert lb.shape == (40, 1) and _logp_err(lb, la.cpu()) < LOGP_TOL
    # adaptive default arguments
    la = sm.log_prob(x0, conditional=c40) if sde_name == "VESDE" else None
    if la is not None:
        assert _logp_err(gm.log_prob(x0, conditional=c40), la.cpu()) < 2e-4
    # Euler-Maruyama with an injected stream, and the public call (same seed, same draws on this device)
    noise = torch.randn(15, B, D_, device=DEV)
    ia, ib = iter(noise), iter(noise)
    sa = sm._sample_sde_from(base, lambda like: next(ia), cond, steps=15)
    sb = gm._sample_sde_from(base, lambda like: next(ib), cond, steps=15)
    assert _state_err(sb, sa.cpu()) < STATE_TOL
    torch.manual_seed(4)
    pa = sm.sample_sde((64, D_), conditional=None if cond is None else cond[:64], steps=10)
    torch.manual_seed(4)
    pb = gm.sample_sde((64, D_), conditional=None if cond is None else cond[:64], steps=10)
    assert _state_err(pb, pa.cpu()) < STATE_TOL
    with pytest.raises(RuntimeError, match="GPU"):
        D.ScoreModel(WrappedMLP(D.MLP(2, 0, 8, [16])), D.VPSDE()).sample_ode_from_base(torch.randn(4, 2), method="euler")


def test_stage_combine_and_normal_fill_entry_points():
    """ff_stage_combine: one pass, zero coefficients never read their array, aliasing allowed, unaligned tails."""
    torch.manual_seed(0)
    for n in (1, 7, 1024, 4099, 1 << 20):
        x = torch.randn(n, device=DEV)
        ks = [torch.randn(n, device=DEV) for _ in range(7)]
        coefs = [0.5, 0.0, -1.25, 0.0, 3.0, 1e-3, -2.0]
        ks[1].fill_(float("nan"))                                   # coefficient 0: must not be read
        exp = x.double()
        for k, c in zip(ks, coefs):
            if c != 0.0:
                exp = exp + c * k.double()
        out = _native.stage_combine(torch.empty_like(x), x, ks, coefs)
        assert (out.double() - exp).abs().max().item() < 1e-5
        # x_coef = 0 (error estimate), views at odd offsets (no 16-byte alignment), out aliasing an input
        big = torch.randn(n + 3, device=DEV)
        xv = big[1:n + 1]
        out = _native.stage_combine(torch.empty_like(xv), xv, [ks[0], None, ks[2]], [2.0, 9.0, 1.0], 0.0)
        assert (out.double() - (2.0 * ks[0].double() + ks[2].double())).abs().max().item() < 1e-5
        y = x.clone()
        _native.stage_combine(y, y, [ks[4]], [0.25])
        assert torch.equal(y, torch.addcmul(x, ks[4], torch.tensor(0.25, device=DEV))) or (y - (x + 0.25 * ks[4])).abs().max() < 1e-6
    with pytest.raises(RuntimeError):
        _native.stage_combine(torch.empty(4), None, [], [])          # CPU tensor


def test_scaled_rms_entry_point_matches_torch_and_is_deterministic():
    """ff_scaled_rms: the adaptive driver's norms in one launch.  Against the torch expressions torchdiffeq uses
    (`_compute_error_ratio`, `_select_initial_step`), NaN / infinity detection, bitwise run-to-run determinism."""
    torch.manual_seed(0)
    atol, rtol = 1e-5, 1e-4
    for n, m in ((1, 1), (37, 5), (1 << 14, 1 << 10), (3_000_001, 50_000)):
        err, y0, y1 = (torch.randn(n, device=DEV) * s for s in (1e-4, 1.0, 1.0))
        le, l0, l1 = (torch.randn(m, device=DEV) * s for s in (1e-3, 3.0, 3.0))
        f0, f1 = torch.randn(n, device=DEV), torch.randn(n, device=DEV)
        got = _native.scaled_rms([(err, None, y0, y1), (le, None, l0, l1), (f1, f0, y0, None)], atol, rtol, check=y1)
        exp = [(err / (atol + rtol * torch.max(y0.abs(), y1.abs()))).double().pow(2).mean().sqrt().item(),
               (le / (atol + rtol * torch.max(l0.abs(), l1.abs()))).double().pow(2).mean().sqrt().item(),
               ((f1 - f0) / (atol + rtol * y0.abs())).double().pow(2).mean().sqrt().item()]
        for g, e in zip(got[:3], exp):
            assert abs(g - e) <= 2e-6 * max(1.0, abs(e)), (n, g, e)
        assert got[3] == 0.0
        again = _native.scaled_rms([(err, None, y0, y1), (le, None, l0, l1), (f1, f0, y0, None)], atol, rtol, check=y1)
        assert again == got                                             # no floating-point atomics
        for bad in (float("nan"), float("inf"), float("-inf")):
            y1b = y1.clone()
            y1b[n // 2] = bad
            assert _native.scaled_rms([(err, None, y0, y1)], atol, rtol, check=y1b)[1] == 1.0
    assert _native.scaled_rms([(torch.zeros(8, device=DEV), None, torch.ones(8, device=DEV), None)], 1.0, 0.0)[0] == 0.0
