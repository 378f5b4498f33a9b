"""The CPU oracle against the golden vectors produced by the reference's own code
(tests/golden/make_golden.py).  These pin every part of the oracle that the reference implements
itself (network, SDE schedules, RHS, divergences, Euler-Maruyama loop, flow dynamics); the ODE
stepper is pinned only in the hybrid sense described in oracle/flowfusion_oracle.py."""
import pytest
import torch

from oracle import flowfusion_oracle as O
from tests._util import flow_oracle, golden_names, load_golden, score_oracle, sde_oracle

# same torch ops in the same order as the reference: agreement to a few ulps (the CPU GEMM's
# summation order depends on thread count and batch shape, so not bit-for-bit)
TIGHT = dict(rtol=1e-5, atol=1e-5)


def close(actual, expected, rtol=1e-5, atol_scale=1e-5):
    torch.testing.assert_close(actual, expected, rtol=rtol, atol=atol_scale * max(1.0, expected.abs().max().item()))


@pytest.mark.parametrize("name", golden_names("mlp_"))
def test_mlp_forward(name):
    meta, a = load_golden(name)
    p = O.mlp_params_from_state_dict(a, "model.")
    cond = a.get("cond")
    torch.testing.assert_close(O.mlp_forward(p, a["t_vec"], a["x"], cond), a["out_vec"], **TIGHT)
    torch.testing.assert_close(O.mlp_forward(p, a["t_scalar"], a["x"], cond), a["out_scalar"], **TIGHT)


def test_sde_schedules():
    _, a = load_golden("sde_schedules")
    t, x = a["t"], a["x"]
    cases = {"vp": O.VP(), "ve": O.VE(), "subvp": O.SubVP(),
             "vp_b": O.VP(beta_min=0.2, beta_max=12.0, T=1.0, epsilon=1e-2),
             "ve_b": O.VE(sigma_min=0.05, sigma_max=25.0, T=1.0, epsilon=1e-4)}
    for key, sde in cases.items():
        torch.testing.assert_close(sde.sigma(t), a[f"{key}_sigma"], rtol=0, atol=0)
        torch.testing.assert_close(sde.diffusion(t, x), a[f"{key}_diffusion"], rtol=0, atol=0)
        torch.testing.assert_close(sde.drift(t, x), a[f"{key}_drift"], rtol=0, atol=0)
        torch.testing.assert_close(sde.epsilon, a[f"{key}_epsilon"], rtol=0, atol=0)
        torch.testing.assert_close(O.normal_log_prob(x, sde.prior_scale()), a[f"{key}_prior_logprob"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", golden_names("score_"))
def test_score_rhs_and_divergences(name):
    meta, a = load_golden(name)
    so = score_oracle(meta, a)
    cond = a.get("cond")
    B = a["x"].shape[0]
    for i in range(3):
        t = a[f"t{i}"]
        close(so.score(t * torch.ones(B), a["x"], cond), a[f"score_{i}"])
        (xdot,) = so.rhs(t, (a["x"],), cond, None)
        close(xdot, a[f"xdot_{i}"])
        xd, div = so.rhs(t, (a["x"], torch.zeros(B, 1)), cond, "hutch", a["e"])
        close(xd, a[f"xdot_hutch_{i}"])
        close(div, a[f"div_hutch_{i}"], 2e-5, 2e-5)
        _, div_e = so.rhs(t, (a["x"], torch.zeros(B, 1)), cond, "exact")
        close(div_e, a[f"div_exact_{i}"], 2e-5, 2e-5)


@pytest.mark.parametrize("name", [n for n in golden_names("sde_") if n != "sde_schedules"])
def test_sample_sde_replay(name):
    meta, a = load_golden(name)
    so = score_oracle(meta, a)
    out = so.sample_sde(a["x_prior"], list(a["noise"]), a.get("cond"), steps=meta["steps"])
    close(out, a["out"], 2e-5, 2e-5)


@pytest.mark.parametrize("name", ["flow_3d", "flow_16d_ragged", "cflow_4d_c2", "cflow_8d_c5"])
def test_flow_dynamics(name):
    meta, a = load_golden(name)
    fo = flow_oracle(a)
    cond = a.get("cond")
    for j in range(3):
        t = a[f"t{j}"]
        torch.testing.assert_close(fo.dynamics(t, a["x"], cond), a[f"v_{j}"], **TIGHT)
        v, div = fo.dynamics_with_jacobian(t, a["x"], cond)
        torch.testing.assert_close(div, a[f"div_{j}"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("name", golden_names("hybrid_score_"))
def test_hybrid_score(name):
    """Oracle RHS + oracle stepper == reference RHS + oracle stepper."""
    meta, a = load_golden(name)
    so = score_oracle(meta, a)
    cond = a.get("cond")
    for run in meta["runs"]:
        m, opts = run["method"], {"step_size": run["step_size"]}
        x0 = so.sample_ode_from_base(a["base"], cond, m, opts)
        scale = a[f"sample_{m}"].abs().max().item()
        torch.testing.assert_close(x0, a[f"sample_{m}"], rtol=2e-5, atol=2e-5 * max(scale, 1.0))
        lp = so.log_prob(a[f"x_data_{m}"], cond, m, opts, "hutch", a[f"e_{m}"])
        torch.testing.assert_close(lp, a[f"lp_hutch_{m}"], rtol=1e-4, atol=1e-3)
        lp_e = so.log_prob(a[f"x_data_{m}"], cond, m, opts, "exact")
        torch.testing.assert_close(lp_e, a[f"lp_exact_{m}"], rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("name", golden_names("hybrid_flow") + golden_names("hybrid_cflow"))
def test_hybrid_flow(name):
    meta, a = load_golden(name)
    fo = flow_oracle(a)
    cond = a.get("cond")
    for run in meta["runs"]:
        m, opts = run["method"], {"step_size": run["step_size"]}
        torch.testing.assert_close(fo.sample(a["xT"], cond, m, opts), a[f"sample_{m}"], rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(fo.log_prob(a[f"x_data_{m}"], cond, m, opts), a[f"logprob_{m}"], rtol=1e-4, atol=1e-4)
