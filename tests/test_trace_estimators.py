"""Hutch++ / XTrace divergence estimators (reference diffusion.py:336-481): CPU tier.

Pinned by fixtures generated from the reference's own ``ScoreModel.forward`` with stored probes
(tests/golden/trace_*.npz: pointwise estimates at three times, and log-densities from the oracle's
fixed-grid stepper driving that forward).  Samples whose probe vectors happen to be linearly dependent
(+-1 entries in 5 dimensions collide often) are left out: there the QR basis, and with it the estimate,
depends on the factorisation's arbitrary completion, in the reference as much as anywhere."""
import pytest
import torch

from flowfusion_amd import _native, adaptive, host_stepper, trace_estimators as TE
from flowfusion_amd.fused import MODE_EXACT
from oracle import flowfusion_oracle as O
from tests import _emulator as E
from tests._util import golden_names, load_golden, score_model, score_oracle

CASES = golden_names("trace_")


def well_posed(P):
    """[B] bool: the probes P[:, b, :] of sample b are linearly independent."""
    return torch.tensor([int(torch.linalg.matrix_rank(P[:, b, :])) == P.shape[0] for b in range(P.shape[1])])


def _close(got, want, mask, tol=2e-5):
    got, want = got.reshape(-1)[mask].double(), want.reshape(-1)[mask].double()
    assert float((got - want).abs().max()) <= tol * max(1.0, float(want.abs().max())), (got, want)


@pytest.mark.parametrize("name", CASES)
def test_oracle_estimators_match_reference(name):
    meta, a = load_golden(name)
    so = score_oracle(meta, a)
    x, cond = a["x"], a.get("cond")
    ok_s, ok_o = well_posed(a["S"]), well_posed(a["O"])
    zero = torch.zeros(x.shape[0], 1)
    for i in range(3):
        t = a[f"t{i}"]
        _close(so.rhs(t, (x, zero), cond, "hutchpp", (a["S"], a["G"]))[1], a[f"div_hpp_{i}"], ok_s, 1e-6)
        _close(so.rhs(t, (x, zero), cond, "xtrace", a["O"])[1], a[f"div_xt_{i}"], ok_o, 1e-6)
    # the hybrid log-densities: same stepper, oracle right-hand side instead of the reference's
    opts = {"step_size": meta["step_size"]}
    times = torch.stack([so.sde.epsilon.to(torch.float32), torch.tensor(1.0)])
    for kind, probes, ok in (("hutchpp", (a["S"], a["G"]), ok_s), ("xtrace", a["O"], ok_o)):
        xT, dlp = O.odeint_fixed(lambda t, y: so.rhs(t, y, cond, kind, probes), (x, zero), times, "rk4", opts)
        lp = dlp + torch.sum(O.normal_log_prob(xT, so.sde.prior_scale()), dim=1, keepdim=True)
        _close(lp, a["lp_hpp_rk4" if kind == "hutchpp" else "lp_xt_rk4"], ok, 2e-6)


@pytest.mark.parametrize("name", CASES)
def test_product_estimators_and_torch_forward_match_reference(name):
    """trace_estimators on a whole Jacobian, and ScoreModel.forward's plain-torch branch, against the fixtures."""
    meta, a = load_golden(name)
    sm = score_model(meta, a, hutchpp=True, hpp_rank=meta["hpp_rank"], hpp_vecs=meta["hpp_vecs"], xt_vecs=meta["xt_vecs"])
    x, cond = a["x"], a.get("cond")
    ok_s, ok_o = well_posed(a["S"]), well_posed(a["O"])
    sm.conditional, sm.prob = cond, True
    sm.S, sm.G, sm.O = a["S"], a["G"], a["O"]
    for i in range(3):
        t = a[f"t{i}"]
        J = torch.autograd.functional.jacobian(lambda v: sm.ode_drift(t, v, conditional=cond).sum(0), x, vectorize=True)
        A = J.permute(1, 2, 0).contiguous()                      # A[b, j, i] = d xdot_i / d x_j
        _close(TE.hutchpp(A, a["S"], a["G"]), a[f"div_hpp_{i}"], ok_s)
        _close(TE.xtrace(A, a["O"]), a[f"div_xt_{i}"], ok_o)
        sm.hutchpp, sm.xtrace = True, False
        xd, div = sm.forward(t.clone(), (x.clone(), torch.zeros(x.shape[0], 1)))
        _close(div.detach(), a[f"div_hpp_{i}"], ok_s)
        torch.testing.assert_close(xd.detach(), a[f"xdot_{i}"], rtol=1e-5, atol=1e-5)
        sm.hutchpp, sm.xtrace = False, True
        _close(sm.forward(t.clone(), (x.clone(), torch.zeros(x.shape[0], 1)))[1].detach(), a[f"div_xt_{i}"], ok_o)


def _cpu_stepper(sm, cond, div_fn):
    net = sm._net()
    plan = _native.plan_words(net.plan(MODE_EXACT))
    wpack = net.wpack("cpu", MODE_EXACT)
    launcher = lambda y, rows, first, count, jac: E.emulate_rhs_jac(plan, wpack, rows, y, cond, first, count, jac)
    return host_stepper.RowStepper(net, "cpu", cond, div_fn, launcher=launcher)


@pytest.mark.parametrize("name", CASES)
def test_row_stepper_reproduces_reference_log_density(name, built_library):
    """Host logic of the estimator solves: the evaluation table run row by row (kernel semantics emulated, whole
    Jacobian per row from unit tangents in the launches the product would use) with the estimators on top gives
    the fixtures' log-densities; the same interpreter as a step function drives the adaptive solver to the oracle's
    dopri5 result."""
    meta, a = load_golden(name)
    sm = score_model(meta, a)
    so = score_oracle(meta, a, torch.float64)
    x, cond = a["x"], a.get("cond")
    eps = float(sm.sde.epsilon)
    table = sm._ode_table(torch.tensor([eps, 1.0]), "rk4", {"step_size": meta["step_size"]}, MODE_EXACT)
    for kind, fn, probes, want in (("hutchpp", lambda A: TE.hutchpp(A, a["S"], a["G"]), (a["S"], a["G"]), a["lp_hpp_rk4"]),
                                   ("xtrace", lambda A: TE.xtrace(A, a["O"]), a["O"], a["lp_xt_rk4"])):
        ok = well_posed(probes[0] if kind == "hutchpp" else probes)
        st = _cpu_stepper(sm, cond, fn)
        xT, dlp = st.run_table(x, table)
        assert st.n_evals == table.shape[0]
        lp = dlp.view(-1, 1) + sm.sde.prior(xT.shape).log_prob(xT).sum(1, keepdim=True)
        _close(lp, want, ok, 3e-5)
    # adaptive: Hutch++ through the Dopri5 driver vs the oracle's dopri5 with the same probes (float64 oracle).
    # (Not for the VE model without sigma normalisation: its random-init right-hand side at t = 1e-5 drives any
    # adaptive controller to vanishing steps.)
    if meta["sde"] == "VESDE":
        return
    # only the well-posed samples take part: the error norm is global, an ill-defined estimate would steer everyone
    ok = well_posed(a["S"])
    xs, cs = x[ok], (None if cond is None else cond[ok])
    Ss, Gs = a["S"][:, ok].contiguous(), a["G"][:, ok].contiguous()
    st = _cpu_stepper(sm, cs, lambda A: TE.hutchpp(A, Ss, Gs))
    solver = adaptive.Dopri5(st.make_step(lambda tr: sm._schedule(tr, "ode")[:3], 1.0), True, 1e-4, 1e-4, None)
    y, lp = solver.integrate(eps, 1.0, xs.clone(), torch.zeros(xs.shape[0]))
    ry, rlp = so.solve_odes_forward(xs.double(), None if cs is None else cs.double(), "dopri5", None, "hutchpp",
                                    (Ss.double(), Gs.double()), atol=1e-4, rtol=1e-4)
    assert solver.n_accepted >= 2
    everyone = torch.ones(xs.shape[0], dtype=torch.bool)
    _close(y, ry.float(), everyone.repeat_interleave(xs.shape[1]), 2e-4)
    _close(lp, rlp, everyone, 2e-4)


def test_thin_qr_follows_lapack():
    """trace_estimators.thin_qr (batched Householder in LAPACK's conventions) against torch.linalg.qr on the CPU
    (LAPACK geqrf + orgqr, what the reference calls): the same Q and R to rounding -- also for rank-deficient sketches
    (two +-1 probes equal up to sign: the reflectors still complete r orthonormal columns), and hence the same
    Hutch++ / XTrace estimates."""
    from flowfusion_amd import trace_estimators as TE
    torch.manual_seed(0)
    for D, r in ((16, 1), (16, 4), (5, 5), (32, 8), (2, 2)):
        Y = torch.randn(50, D, r, dtype=torch.float64)
        Y[3, :, r - 1] = -Y[3, :, 0]                                    # a dependent column
        if D > 2:
            Y[7, 1:, 0] = 0.0                                           # nothing below the diagonal: H = I
        Q, R = TE.thin_qr(Y)
        Qh, Rh = torch.linalg.qr(Y, mode="reduced")
        eye = torch.eye(r, dtype=torch.float64)
        assert (Q.transpose(1, 2) @ Q - eye).abs().max() < 1e-12 and (Q @ R - Y).abs().max() < 1e-12
        assert (torch.tril(R, -1)).abs().max() == 0
        ok = torch.ones(50, dtype=torch.bool)
        ok[3] = r == 1                                                   # the completion of a singular sketch is compared below
        assert (Q[ok] - Qh[ok]).abs().max() < 1e-11 and (R[ok] - Rh[ok]).abs().max() < 1e-11
    A = torch.randn(40, 6, 6, dtype=torch.float64)
    S, G, O = (torch.sign(torch.randn(n, 40, 6, dtype=torch.float64)) for n in (2, 3, 4))
    S[1, 5] = -S[0, 5]                                                   # rank-deficient sketches
    O[3, 9] = O[0, 9]
    own = (TE.hutchpp(A, S, G), TE.xtrace(A, O))
    orig = TE.thin_qr
    try:
        TE.thin_qr = lambda Y: torch.linalg.qr(Y, mode="reduced")
        ref = (TE.hutchpp(A, S, G), TE.xtrace(A, O))
    finally:
        TE.thin_qr = orig
    # Where two probes of a sample coincide up to sign the sketch is singular: the completing column is then a
    # normalised rounding residue -- noise in LAPACK as much as here -- and the estimate is ill-posed.  Everywhere else
    # the two factorisations must give the same numbers.
    def well_posed(P):
        Pc = P.permute(1, 2, 0)
        gram = (Pc.transpose(1, 2) @ Pc).abs() / P.shape[2]
        off = gram - torch.diag_embed(torch.diagonal(gram, 0, 1, 2))
        return off.amax(dim=(1, 2)) < 1.0 - 1e-9
    ws, wo = well_posed(S), well_posed(O)
    assert not ws[5] and not wo[9] and ws.sum() >= 30
    assert (own[0][ws] - ref[0][ws]).abs().max() < 1e-9
    assert (own[1][wo] - ref[1][wo]).abs().max() < 1e-8
    # full-rank request (r = D): whatever the completion, Q is a whole orthonormal basis and Hutch++ IS the trace
    S6 = torch.sign(torch.randn(6, 40, 6, dtype=torch.float64))
    S6[3, 2] = S6[1, 2]
    est = TE.hutchpp(A, S6, G)
    assert (est - torch.diagonal(A, 0, 1, 2).sum(-1)).abs().max() < 1e-9


# ---- the estimators in C (csrc/ff_trace_est.h: the arithmetic of the gfx950 kernel, run on the host) ----------------
@pytest.mark.parametrize("name", CASES)
def test_native_estimators_match_reference_fixtures(name, built_library):
    """ff_trace_estimate_host on the Jacobians of the product's plain-torch drift against the estimates the reference's
    ScoreModel.forward produced with the same probes (diffusion.py:336-481), all three stored times in one call."""
    meta, a = load_golden(name)
    sm = score_model(meta, a)
    x, cond = a["x"], a.get("cond")
    ok_s, ok_o = well_posed(a["S"]), well_posed(a["O"])
    rows = []
    for i in range(3):
        t = a[f"t{i}"]
        J = torch.autograd.functional.jacobian(lambda v: sm.ode_drift(t, v, conditional=cond).sum(0), x, vectorize=True)
        rows.append(J.permute(1, 2, 0).contiguous())              # A[b, j, i] = d xdot_i / d x_j
    A = torch.stack(rows)                                         # [3, B, D, D]
    hpp = _native.trace_estimate(A, "hutchpp", (a["S"], a["G"]), host=True)
    xt = _native.trace_estimate(A, "xtrace", (a["O"],), host=True)
    for i in range(3):
        _close(hpp[i], a[f"div_hpp_{i}"], ok_s)
        _close(xt[i], a[f"div_xt_{i}"], ok_o)


def test_native_estimators_match_the_torch_statement(built_library):
    """... and against trace_estimators.py in float64 on random matrices: every shape class (one probe, several, as many
    as dimensions, more residual probes than dimensions, one dimension), several evaluation rows per call."""
    torch.manual_seed(11)
    for D, r, m, B, n in ((2, 1, 1, 50, 3), (16, 1, 1, 20, 6), (5, 2, 3, 10, 2), (32, 3, 2, 10, 1), (16, 16, 1, 8, 1),
                          (3, 2, 7, 11, 2), (1, 1, 2, 5, 1), (40, 8, 4, 3, 2)):
        A = torch.randn(n, B, D, D)
        S, G = torch.sign(torch.randn(r, B, D)), torch.sign(torch.randn(m, B, D))
        rep = lambda P: P.unsqueeze(1).expand(P.shape[0], n, B, D).reshape(P.shape[0], n * B, D).double()
        ok = well_posed(S).repeat(n)
        flat = A.reshape(n * B, D, D).double()
        scale = max(1.0, float(A.abs().sum(dim=(2, 3)).max()))     # the estimates are sums of D^2 products
        want = TE.hutchpp(flat, rep(S), rep(G))
        got = _native.trace_estimate(A, "hutchpp", (S, G), host=True).reshape(-1).double()
        assert float((got - want)[ok].abs().max()) < 2e-6 * scale, (D, r, m)
        want = TE.xtrace(flat, rep(S))
        got = _native.trace_estimate(A, "xtrace", (S,), host=True).reshape(-1).double()
        assert float((got - want)[ok].abs().max()) < 2e-6 * scale * max(1, r), (D, r)
        if r == D:                                                 # the sketch spans everything: Hutch++ IS the trace
            tr = torch.diagonal(A, 0, 2, 3).sum(-1).reshape(-1).double()
            est = _native.trace_estimate(A, "hutchpp", (S, G), host=True).reshape(-1).double()
            assert float((est - tr)[ok].abs().max()) < 2e-6 * scale
    # a column with nothing below the diagonal (H = I) and a rank-deficient sketch at full rank request
    A = torch.randn(1, 4, 3, 3)
    A[0, 0, 1:, :] = 0.0
    S = torch.sign(torch.randn(3, 4, 3))
    S[2, 1] = -S[0, 1]
    G = torch.sign(torch.randn(1, 4, 3))
    est = _native.trace_estimate(A, "hutchpp", (S, G), host=True)[0]
    assert torch.isfinite(est).all()
    with pytest.raises(RuntimeError, match="do not match"):
        _native.trace_estimate(A, "hutchpp", (S[:, :2], G), host=True)
    with pytest.raises(ValueError, match="expected 'hutchpp' or 'xtrace'"):
        _native.trace_estimate(A, "hutch", (S,), host=True)
