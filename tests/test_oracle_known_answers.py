"""Known-answer tests that do not depend on any stepper implementation other than the one under
test: convergence order of the restated fixed-grid schemes, time reversal, the grid rule, and
closed-form Gaussian densities pushed through the probability-flow ODE."""
import math

import pytest
import torch

from oracle import flowfusion_oracle as O


def _solve_linear(method, n, t0=0.0, t1=1.0, dtype=torch.float64):
    # dy/dt = -2 t y + cos(3t) has a smooth non-polynomial solution; compare two resolutions with a fine one
    f = lambda t, y: (-2.0 * t * y[0] + torch.cos(3.0 * t),)
    y0 = (torch.tensor([1.0, -0.5], dtype=dtype),)
    t = torch.tensor([t0, t1], dtype=dtype)
    (y,) = O.odeint_fixed(f, y0, t, method, {"step_size": abs(t1 - t0) / n})
    return y


@pytest.mark.parametrize("method,order", [("euler", 1), ("midpoint", 2), ("heun3", 3), ("rk4", 4),
                                          ("rk4_classic", 4), ("dopri5_fixed", 5)])
def test_convergence_order(method, order):
    ref = _solve_linear("dopri5_fixed", 4000)
    e1 = (_solve_linear(method, 20) - ref).abs().max().item()
    e2 = (_solve_linear(method, 40) - ref).abs().max().item()
    observed = math.log2(e1 / e2)
    assert abs(observed - order) < 0.35, (method, observed)


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_reversed_span_equals_negated_problem(method):
    """Solving t: 1 -> 0 must equal solving s = -t: -1 -> 0 of the negated field (how torchdiffeq does it)."""
    f = lambda t, y: (torch.sin(t) * y[0] + t,)
    y0 = (torch.tensor([0.3, 1.2], dtype=torch.float64),)
    (a,) = O.odeint_fixed(f, y0, torch.tensor([1.0, 0.0], dtype=torch.float64), method, {"step_size": 0.05})
    g = lambda s, y: (-(torch.sin(-s) * y[0] + (-s)),)
    (b,) = O.odeint_fixed(g, y0, torch.tensor([-1.0, 0.0], dtype=torch.float64), method, {"step_size": 0.05})
    torch.testing.assert_close(a, b, rtol=0, atol=0)


def test_grid_rule():
    t = torch.tensor([0.0, 1.0])
    g = O.grid_from_step_size(t, 0.25)
    torch.testing.assert_close(g, torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0]))
    g = O.grid_from_step_size(t, 0.3)       # last interval is cut short at t1
    torch.testing.assert_close(g, torch.tensor([0.0, 0.3, 0.6, 0.9, 1.0]))
    # no step_size: the grid is t itself -> a single step
    f = lambda tt, y: (y[0],)
    (y,) = O.odeint_fixed(f, (torch.ones(1, dtype=torch.float64),), torch.tensor([0.0, 1.0], dtype=torch.float64), "euler", None)
    assert y.item() == 2.0


class _GaussianScore(O.ScoreOracle):
    """Analytic score of data ~ N(mu, s^2 I) diffused by the SDE: the marginal at time t is
    N(m_t mu, (m_t s)^2 + sigma_t^2)."""

    def __init__(self, sde, mu, s, dtype=torch.float64):
        self.sde, self.mu, self.s, self.dtype, self.no_sigma = sde, mu, s, dtype, True

    def moments(self, t):
        if self.sde.kind == "ve":
            m, sig = torch.ones_like(t), self.sde.sigma(t)
        else:
            m, sig = self.sde.marginal_prob_scalars(t)
        return m, (m * self.s) ** 2 + sig ** 2

    def score(self, t, x, conditional=None):
        t = t * torch.ones(x.shape[0], dtype=x.dtype) if t.dim() == 0 else t
        m, var = self.moments(t)
        return -(x - m[:, None] * self.mu) / var[:, None]


def _gauss_logpdf(x, mean, var):
    d = x.shape[1]
    return -0.5 * ((x - mean) ** 2).sum(1) / var - 0.5 * d * torch.log(2 * math.pi * var)


@pytest.mark.parametrize("sde_name", ["vp", "ve", "subvp"])
def test_gaussian_density_through_pf_ode(sde_name):
    dt = torch.float64
    sde = {"vp": O.VP(dtype=dt), "ve": O.VE(epsilon=1e-3, dtype=dt), "subvp": O.SubVP(dtype=dt)}[sde_name]
    D = 3
    mu = torch.tensor([0.5, -1.0, 2.0], dtype=dt)
    s = 0.6
    so = _GaussianScore(sde, mu, s)
    torch.manual_seed(0)
    x0 = mu + s * torch.randn(64, D, dtype=dt)
    eps = sde.epsilon
    errs = []
    for n in (40, 80):
        opts = {"step_size": float((1.0 - eps) / n)}
        xT, dlp = so.solve_odes_forward(x0, None, "rk4", opts, divergence="exact")
        m1, v1 = so.moments(torch.tensor([1.0], dtype=dt))
        me, ve = so.moments(eps.reshape(1))
        lhs = _gauss_logpdf(xT, m1 * mu, v1) + dlp.squeeze(1)      # log p_1(x(1)) + int div
        rhs = _gauss_logpdf(x0, me * mu, ve)                        # log p_eps(x(eps))
        errs.append((lhs - rhs).abs().max().item())
    assert errs[1] < 2e-4, errs
    assert errs[1] < errs[0] / 6.0, errs        # ~4th order (16x per halving; loose bound)


def test_gaussian_sampling_transport():
    """PF-ODE transport of a Gaussian is affine and known in closed form: x(eps) = m_e mu + sqrt(v_e/v_1) (x(1) - m_1 mu)."""
    dt = torch.float64
    sde = O.VP(dtype=dt)
    mu = torch.tensor([1.0, -2.0], dtype=dt)
    so = _GaussianScore(sde, mu, 0.8)
    torch.manual_seed(1)
    z = torch.randn(32, 2, dtype=dt)
    x = so.sample_ode_from_base(z, None, "rk4", {"step_size": float((1.0 - sde.epsilon) / 200)})
    m1, v1 = so.moments(torch.tensor([1.0], dtype=dt))
    me, ve = so.moments(sde.epsilon.reshape(1))
    expect = me * mu + torch.sqrt(ve / v1) * (z - m1 * mu)
    torch.testing.assert_close(x, expect, rtol=1e-6, atol=1e-6)


# ---- the embedded pairs' tableaux against an independent published source, and against their order conditions -----------
def _order_residuals(c, A, b, order):
    """Largest violation of the Runge-Kutta order conditions up to `order` (rooted trees up to 4 nodes spelled out, plus the
    quadrature conditions sum_i b_i c_i^k = 1/(k+1) up to order-1) for nodes c, matrix A and weights b."""
    import numpy as np
    c, A, b = np.asarray(c, float), np.asarray(A, float), np.asarray(b, float)
    res = [abs(b.sum() - 1.0)]
    for k in range(1, order):
        res.append(abs((b * c ** k).sum() - 1.0 / (k + 1)))
    if order >= 3:
        res.append(abs(b @ A @ c - 1.0 / 6))
    if order >= 4:
        res += [abs((b * c) @ A @ c - 1.0 / 8), abs(b @ A @ (c ** 2) - 1.0 / 12), abs(b @ A @ A @ c - 1.0 / 24)]
    res.append(np.abs(A.sum(axis=1) - c).max())          # row sums
    return max(res)


def _as_matrix(alpha, beta, stages):
    import numpy as np
    c = np.zeros(stages)
    A = np.zeros((stages, stages))
    for i, (al, row) in enumerate(zip(alpha, beta)):
        c[i + 1] = al
        A[i + 1, : len(row)] = row
    return c, A


def test_embedded_tableaux_against_scipy_and_order_conditions():
    """torchdiffeq is not available offline (the steppers' parity is unpinned), but the NUMBERS of two of
    its pairs are published elsewhere in this image: scipy's RK45 is Dormand-Prince 5(4) (same nodes, matrix and fifth-order
    weights; its error estimate uses Dormand-Prince's own fourth-order weights where torchdiffeq uses Shampine's, so c_error is
    not comparable) and RK23 is Bogacki-Shampine 3(2) (everything comparable, error up to sign).  What scipy cannot pin --
    torchdiffeq's c_error of dopri5, fehlberg2, adaptive_heun -- must at least be what it claims to be: c_sol of the stated
    order, c_sol - c_error an embedded solution one order lower."""
    import numpy as np
    from scipy.integrate._ivp import rk
    from flowfusion_amd import adaptive, solvers
    from oracle import flowfusion_oracle as O
    # product, oracle and the fixed-step Dormand-Prince table all carry the same numbers
    assert list(adaptive.ALPHA) == list(O._DP5_ALPHA) and [list(r) for r in adaptive.BETA] == [list(r) for r in O._DP5_BETA]
    assert list(adaptive.C_SOL) == list(O._DP5_C_SOL) and list(adaptive.C_ERROR) == list(O._DP5_C_ERR)
    c, A = _as_matrix(adaptive.ALPHA, adaptive.BETA, 7)
    np.testing.assert_allclose(c[:6], rk.RK45.C, rtol=0, atol=1e-16)
    np.testing.assert_allclose(A[:6, :5], rk.RK45.A, rtol=0, atol=2e-16)
    np.testing.assert_allclose(adaptive.C_SOL[:6], rk.RK45.B, rtol=0, atol=1e-16)
    np.testing.assert_allclose(A[6, :6], rk.RK45.B, rtol=0, atol=1e-16)                      # first same as last
    dp = solvers.DOPRI5_FIXED
    np.testing.assert_allclose([0.0 if v is None else v for v in dp.c[:5]], rk.RK45.C[:5], atol=1e-16)
    np.testing.assert_allclose(dp.b, rk.RK45.B, atol=1e-16)
    bs = adaptive.TABLEAUX["bosh3"]
    cb, Ab = _as_matrix(bs.alpha, bs.beta, 4)
    np.testing.assert_allclose(cb[:3], rk.RK23.C, atol=1e-16)
    np.testing.assert_allclose(Ab[:3, :3], rk.RK23.A, atol=1e-16)
    np.testing.assert_allclose(bs.c_sol[:3], rk.RK23.B, atol=1e-16)
    np.testing.assert_allclose(np.asarray(bs.c_error), -rk.RK23.E, atol=1e-16)                # error up to sign
    # order conditions: c_sol of order p, c_sol - c_error of order p - 1 (the pair's `order` is p)
    for name, tab in adaptive.TABLEAUX.items():
        c, A = _as_matrix(tab.alpha, tab.beta, tab.stages)
        hi = _order_residuals(c, A, tab.c_sol, min(tab.order, 4))
        lo = _order_residuals(c, A, np.asarray(tab.c_sol) - np.asarray(tab.c_error), min(tab.order - 1, 4))
        assert hi < 1e-15 and lo < 1e-15, (name, hi, lo)
    # ... and for dopri5 the quadrature conditions up to the full orders 5 and 4 (bushy trees)
    c, A = _as_matrix(adaptive.ALPHA, adaptive.BETA, 7)
    for k in range(5):
        assert abs((np.asarray(adaptive.C_SOL) * c ** k).sum() - 1 / (k + 1)) < 1e-15
    for k in range(4):
        assert abs(((np.asarray(adaptive.C_SOL) - np.asarray(adaptive.C_ERROR)) * c ** k).sum() - 1 / (k + 1)) < 1e-15
    # dense output: torchdiffeq's c_mid reproduces y(t0 + dt/2) to fifth order for y' = y-like polynomials: weights sum to 1/2
    assert abs(sum(adaptive.C_MID) - 0.5) < 1e-15
    for k in range(1, 4):
        assert abs((np.asarray(adaptive.C_MID) * c ** k).sum() - 0.5 ** (k + 1) / (k + 1)) < 1e-12, k


def test_fixed_grid_tableaux_satisfy_their_order_conditions():
    """euler (1), midpoint (2), heun3 (3), rk4 = the 3/8 rule and rk4_classic (4), fixed-step Dormand-Prince (5: checked to
    4 with all trees and to 5 on the quadrature conditions): the product's evaluation-row programs are built from these."""
    import numpy as np
    from flowfusion_amd import solvers
    want = {"euler": 1, "midpoint": 2, "heun3": 3, "rk4": 4, "rk4_classic": 4, "dopri5_fixed": 5}
    for name, tab in solvers.FIXED_METHODS.items():
        S = tab.stages
        c = np.array([1.0 if v is None else v for v in tab.c], float)
        A = np.zeros((S, S))
        for i, row in enumerate(tab.A):
            A[i, : len(row)] = row
        assert _order_residuals(c, A, tab.b, min(want[name], 4)) < 1e-15, name
        for k in range(want[name]):
            assert abs((np.asarray(tab.b) * c ** k).sum() - 1 / (k + 1)) < 1e-15, (name, k)
    # the 3/8 rule is what torchdiffeq calls rk4 (rk4_alt_step_func): its nodes are thirds, not halves
    assert list(solvers.RK4_38.c[:3]) == [0.0, 1 / 3, 2 / 3] and list(solvers.RK4_38.b) == [0.125, 0.375, 0.375, 0.125]


def test_dopri8_tableau_is_what_it_claims_to_be():
    """Dormand-Prince 8(7)13M restated from memory of torchdiffeq's dopri8.py (not available offline) -- verified, not
    trusted: row sums, all rooted trees to order 4, the quadrature conditions to k = 7 for the eighth-order weights and to
    k = 6 for the embedded seventh-order ones (which must FAIL k = 7: they are not an eighth-order formula), observed
    convergence order 8 / 7 on a nonlinear scalar ODE; the build's own midpoint weights satisfy every order condition up to 5
    at theta = 1/2 and the quadrature conditions to k = 7.  Product and oracle carry the same numbers."""
    import numpy as np
    from flowfusion_amd import adaptive
    from oracle import flowfusion_oracle as O
    tab = adaptive.WIDE_TABLEAUX["dopri8"]
    order, o_alpha, o_beta, o_sol, o_err, o_mid = O._ADAPTIVE_TABLEAUX["dopri8"]
    assert (order, tab.order, tab.stages) == (8, 8, 14)
    assert list(tab.alpha) == list(o_alpha) and [list(r) for r in tab.beta] == [list(r) for r in o_beta]
    assert list(tab.c_sol) == list(o_sol) and list(tab.c_error) == list(o_err) and list(tab.c_mid) == list(o_mid)
    c, A = _as_matrix(tab.alpha, tab.beta, 14)
    b8 = np.asarray(tab.c_sol)
    b7 = b8 - np.asarray(tab.c_error)
    assert np.abs(A.sum(axis=1) - c).max() < 4e-15
    assert _order_residuals(c, A, b8, 4) < 4e-15 and _order_residuals(c, A, b7, 4) < 4e-15
    for k in range(8):
        assert abs((b8 * c ** k).sum() - 1 / (k + 1)) < 1e-15, k
    for k in range(7):
        assert abs((b7 * c ** k).sum() - 1 / (k + 1)) < 1e-15, k
    assert abs((b7 * c ** 7).sum() - 1 / 8) > 1e-5
    assert np.array_equal(A[13, :13], b8[:13]) and b8[13] == 0.0                  # first same as last
    # midpoint weights: rooted trees up to order 5 at theta = 1/2, quadrature to k = 7
    m, th = np.asarray(tab.c_mid), 0.5
    Ac, Ac2, Ac3 = A @ c, A @ c ** 2, A @ c ** 3
    AAc = A @ Ac
    trees = [(np.ones(14), th), (c, th ** 2 / 2), (c ** 2, th ** 3 / 3), (Ac, th ** 3 / 6), (c ** 3, th ** 4 / 4),
             (c * Ac, th ** 4 / 8), (Ac2, th ** 4 / 12), (AAc, th ** 4 / 24), (c ** 4, th ** 5 / 5), (c ** 2 * Ac, th ** 5 / 10),
             (c * Ac2, th ** 5 / 15), (c * AAc, th ** 5 / 30), (Ac * Ac, th ** 5 / 20), (Ac3, th ** 5 / 20),
             (A @ (c * Ac), th ** 5 / 40), (A @ Ac2, th ** 5 / 60), (A @ AAc, th ** 5 / 120)]
    trees += [(c ** k, th ** (k + 1) / (k + 1)) for k in (5, 6, 7)]
    assert max(abs(m @ phi - want) for phi, want in trees) < 1e-15
    # observed order on y' = cos(t) y (y = exp(sin t)): halving the step divides the error by ~2^8 (b8) and ~2^7 (b7)
    f = lambda t, y: np.cos(t) * y

    def solve(b, n):
        y, t, h = 1.0, 0.0, 2.0 / n
        for _ in range(n):
            k = []
            for i in range(13):
                k.append(f(t + c[i] * h, y + h * sum(A[i, j] * k[j] for j in range(i))))
            y, t = y + h * sum(b[i] * k[i] for i in range(13)), t + h
        return abs(y - np.exp(np.sin(2.0)))
    assert 7.5 < np.log2(solve(b8, 4) / solve(b8, 8)) < 8.8 and 7.5 < np.log2(solve(b8, 8) / solve(b8, 16)) < 8.8
    assert 6.5 < np.log2(solve(b7, 4) / solve(b7, 8)) < 9.0
    assert solve(b8, 16) < 1e-12


@pytest.mark.parametrize("method", ["dopri5", "bosh3"])
def test_step_control_law_against_scipy(method):
    """An executable third-party anchor for the part of the adaptive solver the tableau pins cannot reach.  torchdiffeq's
    step control (`_select_initial_step`, the RMS error ratio against atol + rtol max(|y0|, |y1|), safety 0.9, growth at most
    10) is Hairer's, and scipy's RK45 -- in this image -- implements the same rules for the same Dormand-Prince pair
    (torchdiffeq's rk_common.py cites scipy for the initial step).  They differ in four places: the error WEIGHTS
    (torchdiffeq's `c_error` is built from another fourth-order companion -- exactly 2/3 of the published b - b^ that scipy
    holds, asserted below; the tableau test pins them by their order conditions), what happens AFTER a rejected step (scipy
    caps the next growth factor at 1), after an accepted step with an error ratio above 0.9^order (scipy shrinks, torchdiffeq
    keeps the step) and the end of the span (scipy clips the last step, torchdiffeq interpolates).  So
    scipy's stepper is given the oracle's error weights, and on a smooth problem in float64 the oracle's first step and
    every step of its rejection-free prefix must then be scipy's, to rounding; a wrong exponent, norm, safety factor or
    initial-step rule shows at once."""
    import numpy as np
    from scipy.integrate import RK23, RK45
    rng = np.random.default_rng(0)
    A = rng.normal(size=(6, 6)) * 0.4 - 0.8 * np.eye(6)

    def rhs_np(t, y):
        return A @ y + np.sin(3.0 * t + y) * 0.5

    if method == "dopri5":
        Base, c_err, order = RK45, np.array([float(c) for c in O._DP5_C_ERR]), 5
        assert np.allclose(c_err, -2.0 / 3.0 * RK45.E, rtol=0, atol=1e-15)      # same direction, 2/3 of the published estimate
    else:       # Bogacki-Shampine 3(2) = scipy's RK23: the error weights are the published ones (up to sign)
        Base, order = RK23, 3
        c_err = np.array([float(c) for c in O._ADAPTIVE_TABLEAUX["bosh3"][4]])
        assert O._ADAPTIVE_TABLEAUX["bosh3"][0] == 3 and np.allclose(np.abs(c_err), np.abs(RK23.E), rtol=0, atol=1e-15)

    class RK45WithOracleWeights(Base):
        E = c_err

    y0 = rng.normal(size=6)
    total = 0
    for rtol, atol in ((1e-6, 1e-8), (1e-4, 1e-6), (1e-8, 1e-10), (1e-3, 1e-3), (1e-10, 1e-12)):
        sp = RK45WithOracleWeights(rhs_np, 0.0, y0, t_bound=100.0, rtol=rtol, atol=atol)
        first_h = sp.h_abs
        sp_steps = []
        for _ in range(40):
            sp.step()
            sp_steps.append((sp.t_old, sp.t - sp.t_old))
        At = torch.tensor(A)
        func = lambda t, y: (y[0] @ At.T + torch.sin(3.0 * t + y[0]) * 0.5,)
        O.odeint_dopri5(func, (torch.tensor(y0)[None, :],), torch.tensor([0.0, 4.0], dtype=torch.float64), rtol=rtol, atol=atol,
                        method=method)
        steps = O.last_adaptive_stats["steps"]
        assert abs(steps[0][1] - first_h) <= 1e-12 * first_h                      # the initial-step rule
        # accepted steps up to and including the first one after the first rejection (the retry follows the same rule in
        # both; only the step after it may differ: scipy caps its growth at 1)
        # ... and torchdiffeq never shrinks the step after an ACCEPTED one (its `dfactor` is 1 below an error ratio of 1),
        # where scipy applies 0.9 ratio^-1/5 whatever it is: the walk also ends behind an accepted ratio above 0.9^5
        accepted, seen_reject = [], False
        for t, dt, ratio, ok in steps:
            if ok:
                accepted.append((t, dt))
                if seen_reject or ratio > 0.9 ** order:
                    break
            else:
                seen_reject = True
        assert len(accepted) >= 2, (rtol, steps[:6])
        for n, ((t, dt), (ts, hs)) in enumerate(zip(accepted, sp_steps)):
            assert abs(t - ts) <= 1e-9 * max(1.0, abs(ts)) and abs(dt - hs) <= 1e-7 * hs, (rtol, n, t, dt, ts, hs)
        total += len(accepted)
    assert total >= 12, total


def _closed_form_step(f):
    """A step function with FusedNet.make_step's contract for a closed-form right-hand side, as adaptive.HostSteppedPair
    uses it: one evaluation row, the derivative back as aux_0."""
    return lambda y, k1, lp0, kl1, t_rows, cin, slots, tail, use_y, n_aux: (f(t_rows[0], y)[None], None)


def test_dopri8_dense_output_does_not_lean_on_the_restated_midpoint():
    """The dopri8 dense-output MIDPOINT weights are this build's own derivation (torchdiffeq's could not be restated,
    DESIGN.md section 6), and product and oracle share them -- so two checks that do not:
    (1) a last step that lands exactly on t_end (x == 1): torchdiffeq evaluates its quartic there too (no shortcut), whose
        midpoint terms cancel -- the answer must be the step's y1 to a few ulp WHATEVER the midpoint was;
    (2) t_end strictly inside the last step of a closed-form problem: the answer must sit within the solver tolerance of
        the closed form (a wrong midpoint would show as an O(dt^4) interpolation error)."""
    from flowfusion_amd import adaptive
    rhs = lambda t, y: -(t * y)                                     # y(t) = y0 exp(-t^2 / 2)
    y0 = torch.tensor([[1.0, -2.5, 0.3]])
    exact = lambda t: y0.double() * math.exp(-t * t / 2)
    opts = {"first_step": 0.4, "max_step": 0.4}                     # 0.4 + 0.4 == 0.8 exactly in float64
    s = adaptive.HostSteppedPair(_closed_form_step(rhs), False, 1e-6, 1e-6, dict(opts), method="dopri8")
    y_end, _ = s.integrate(0.0, 0.8, y0.clone(), None)
    assert s.n_accepted == 2
    ulp = 2.0 ** -23 * float(y0.abs().max())
    # two 8th-order steps of 0.4 are far below fp32 rounding; what is left is the quartic itself, whose terms (18 y0, 32 y_mid,
    # ...) are rounded at ~32 |y| ulp -- torchdiffeq's fp32 evaluation loses the same
    assert float((y_end.double() - exact(0.8)).abs().max()) < 64 * ulp
    tab = adaptive.WIDE_TABLEAUX["dopri8"]
    broken = adaptive.EmbeddedTableau(tab.name, tab.order, tab.alpha, tab.beta, tab.c_sol, tab.c_error,
                                      tuple(0.0 for _ in tab.c_mid))            # a midpoint that is simply y0
    s2 = adaptive.HostSteppedPair(_closed_form_step(rhs), False, 1e-6, 1e-6, dict(opts), method="dopri8")
    s2.tab = broken
    y_broken, _ = s2.integrate(0.0, 0.8, y0.clone(), None)
    assert float((y_broken - y_end).abs().max()) <= 64 * ulp         # x == 1: the midpoint's 16 - 32 + 16 cancel to rounding
    # (2) inside the last step: steps of 0.1 up to 1.3, t_end = 1.27 (x = 0.7)
    s3 = adaptive.HostSteppedPair(_closed_form_step(rhs), False, 1e-6, 1e-6, {"first_step": 0.1, "max_step": 0.1}, method="dopri8")
    y_in, _ = s3.integrate(0.0, 1.27, y0.clone(), None)
    assert s3.n_accepted == 13
    err = float((y_in.double() - exact(1.27)).abs().max())
    assert err < 2e-6, err
    # ... and the check has teeth: the broken midpoint misses the same bar by orders of magnitude
    s4 = adaptive.HostSteppedPair(_closed_form_step(rhs), False, 1e-6, 1e-6, {"first_step": 0.1, "max_step": 0.1}, method="dopri8")
    s4.tab = broken
    y_bad, _ = s4.integrate(0.0, 1.27, y0.clone(), None)
    assert float((y_bad.double() - exact(1.27)).abs().max()) > 100 * max(err, 1e-7)


def test_oracle_adaptive_solves_converge_to_scipys_integration():
    """The oracle's restated adaptive steppers against an integrator nobody here wrote: scipy's `solve_ivp` (RK45, rtol 1e-11)
    on the same float64 right-hand side -- a random-init score model's probability-flow ODE, decreasing span, and its
    exact-trace log-density.  Every pair must land within a small multiple of its tolerance of the converged answer."""
    import numpy as np
    from scipy.integrate import solve_ivp
    torch.manual_seed(21)
    D, B = 3, 6
    params = O.MLPParams(W=torch.randn(4) * 16, pi=torch.tensor(math.pi),
                         weights=[torch.randn(32, 8 + D) * 0.3, torch.randn(32, 32) * 0.2, torch.randn(D, 32) * 0.2],
                         biases=[torch.randn(32) * 0.1, torch.randn(32) * 0.1, torch.randn(D) * 0.1])
    so = O.ScoreOracle(params, O.VE(dtype=torch.float64), no_sigma=False, dtype=torch.float64)
    eps = float(so.sde.epsilon.to(torch.float32))
    base = torch.randn(B, D, dtype=torch.float64)

    def f(t, y):
        (xd,) = so.rhs(torch.tensor(t, dtype=torch.float64), (torch.from_numpy(y.reshape(B, D).copy()),), None, None)
        return xd.reshape(-1).numpy()
    conv = solve_ivp(f, (1.0, eps), (base * so.sde.base_scale).reshape(-1).numpy(), method="RK45", rtol=1e-11, atol=1e-13)
    assert conv.success
    want = torch.from_numpy(conv.y[:, -1].reshape(B, D))
    scale = float(want.abs().max())
    for method, tol, bar in (("dopri5", 1e-8, 1e-6), ("dopri5", 1e-5, 1e-3), ("bosh3", 1e-6, 1e-4), ("fehlberg2", 1e-5, 1e-3),
                             ("adaptive_heun", 1e-5, 1e-3), ("dopri8", 1e-9, 1e-7)):
        got = so.sample_ode_from_base(base, None, method, None, tol, tol)
        assert float((got - want).abs().max()) / scale < bar, (method, tol, float((got - want).abs().max()) / scale)

    x0 = torch.randn(B, D, dtype=torch.float64) * 0.5

    def g(t, y):
        x = torch.from_numpy(y[: B * D].reshape(B, D).copy())
        xd, div = so.rhs(torch.tensor(t, dtype=torch.float64), (x, torch.zeros(B, 1, dtype=torch.float64)), None, "exact", None)
        return np.concatenate([xd.detach().reshape(-1).numpy(), div.detach().reshape(-1).numpy()])
    conv = solve_ivp(g, (eps, 1.0), np.concatenate([x0.reshape(-1).numpy(), np.zeros(B)]), method="RK45", rtol=1e-11, atol=1e-13)
    assert conv.success
    xT = torch.from_numpy(conv.y[: B * D, -1].reshape(B, D))
    want_lp = torch.from_numpy(conv.y[B * D:, -1]).reshape(B, 1) + O.normal_log_prob(xT, so.sde.prior_scale()).sum(1, keepdim=True)
    for method, tol, bar in (("dopri5", 1e-8, 1e-6), ("bosh3", 1e-6, 1e-4)):
        lp = so.log_prob(x0, None, method, None, "exact", None, tol, tol)
        assert float(((lp - want_lp).abs() / want_lp.abs().clamp_min(1.0)).max()) < bar, (method, tol)
