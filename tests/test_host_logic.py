"""CPU tier: host-side logic of the product (no GPU compute).

* the C-ABI library loads here and exports every symbol include/flowfusion_amd.h declares;
* kernel plan selection and argument checking;
* the weight packer + evaluation tables, checked end-to-end through a CPU emulation of the kernel's
  semantics against the oracle (tests/_emulator.py);
* state_dict compatibility with reference checkpoints (the golden fixtures hold reference
  state_dicts) and the plain-torch pointwise members against the golden vectors;
* error behaviour: unsupported requests raise, nothing falls back to the CPU.
"""
import ctypes
import re
from pathlib import Path

import pytest
import torch

from flowfusion_amd import _native, solvers
from flowfusion_amd import diffusion as D
from flowfusion_amd import flow as F
from flowfusion_amd.fused import MODE_EXACT, MODE_HUTCH, MODE_STATE
from oracle import flowfusion_oracle as O
from tests import _emulator as E
from tests._util import (flow_model, flow_oracle, golden_names, load_golden, max_rel, score_model, score_oracle)

ROOT = Path(__file__).resolve().parents[1]


# ---- C ABI ---------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol(built_library):
    header = (ROOT / "include" / "flowfusion_amd.h").read_text()
    declared = set(re.findall(r"\b(ff_[a-z_0-9]+)\s*\(", header))
    assert {"ff_mlp_plan", "ff_mlp_wpack", "ff_mlp_ode_launch", "ff_version"} <= declared
    for sym in sorted(declared):
        assert hasattr(built_library, sym), f"libflowfusion_amd.so does not export {sym}"
    assert b"gfx950" in built_library.ff_version()
    assert built_library.ff_kernel_count() >= 1
    assert built_library.ff_kernel_name(0) is not None


def test_plan_selection_and_errors(built_library):
    p = _native.make_plan(16, 0, [256] * 4, MODE_STATE)          # bench shape: 16x16x4, two waves per SIMD
    assert (p.tile, p.width, p.dregs, p.cregs) == (16, 256, 4, 0)
    p = _native.make_plan(2, 0, [128] * 3, MODE_HUTCH)            # the notebooks' 3x128: 16x16x4, two waves per SIMD (round 3)
    assert (p.tile, p.width, p.dregs) == (16, 128, 4)
    p = _native.make_plan(2, 0, [64] * 3, MODE_HUTCH)             # narrower nets, and 128-wide ones beyond 16 dimensions: 32x32x2
    assert (p.tile, p.width, p.dregs) == (32, 64, 4)
    p = _native.make_plan(20, 0, [128] * 3, MODE_STATE)
    assert (p.tile, p.width, p.dregs) == (32, 128, 16)
    p = _native.make_plan(32, 8, [200, 100], MODE_STATE)          # ragged widths pad to the max
    assert (p.width, p.cond_dim) == (256, 8) and p.dregs * (64 // p.tile) >= 32
    p = _native.make_plan(48, 0, [256, 256], MODE_EXACT)          # many dimensions on a 256-wide net stay 256 wide
    assert (p.tile, p.width, p.dregs) == (16, 256, 16)
    p = _native.make_plan(6, 24, [128, 128], MODE_HUTCH)          # more than 16 conditional inputs: the 256-wide catch-all
    assert (p.tile, p.width, p.cregs) == (16, 256, 8)
    p = _native.make_plan(64, 0, [512] * 5, MODE_STATE)           # BASELINE config 4: 16x16x4 kernels
    assert (p.tile, p.width, p.dregs) == (16, 512, 16)
    p = _native.make_plan(64, 0, [512] * 5, MODE_EXACT)           # exact trace: passes of tile-1 tangents
    assert p.tile == 16 and _native.samples_per_workgroup(p, MODE_EXACT) == 4
    with pytest.raises(NotImplementedError):
        _native.make_plan(64, 0, [2048] * 2, MODE_STATE)          # wider than any compiled kernel
    with pytest.raises(NotImplementedError):
        _native.make_plan(130, 0, [64], MODE_STATE)               # more dimensions than any compiled kernel
    with pytest.raises(NotImplementedError):
        _native.make_plan(4, 65, [64], MODE_STATE)                # more conditional inputs than any compiled kernel
    # beyond 512 wide / 64 dimensions / 32 conditional inputs: the wide catch-alls (a tile per workgroup at every batch size)
    p = _native.make_plan(64, 0, [1024] * 2, MODE_STATE)
    assert (p.tile, p.width, p.dregs, p.cregs) == (16, 1024, 32, 16) and _native.kernel_name(p).endswith("_wide")
    assert _native.samples_per_workgroup(p, MODE_STATE) == 16
    p = _native.make_plan(100, 40, [300], MODE_HUTCH)
    assert (p.width, p.dregs, p.cregs) == (1024, 32, 16) and _native.samples_per_workgroup(p, MODE_HUTCH) == 8
    # launch argument checking happens before any HIP call
    bad = _native.OdeArgs()
    rc = built_library.ff_mlp_ode_launch(ctypes.byref(p), ctypes.byref(bad), None)
    assert rc == _native.FF_ERR_BADARG


def test_no_cpu_fallback(built_library):
    sm = D.ScoreModel(D.MLP(4, 0, 8, [64, 64]), D.VPSDE(), no_sigma=True).eval()
    with pytest.raises(RuntimeError, match="GPU"):
        sm.sample_ode_from_base(torch.randn(8, 4), method="rk4", options={"step_size": 0.1})
    with pytest.raises(RuntimeError, match="GPU"):
        sm.sample_ode_from_base(torch.randn(8, 4))                # reference default method=dopri5 (adaptive)
    with pytest.raises(RuntimeError, match="GPU"):
        sm.sample_ode_from_base(torch.randn(8, 4), method="dopri8")          # stage-by-stage driver: GPU only as well
    with pytest.raises(NotImplementedError, match="multistep"):
        sm.sample_ode_from_base(torch.randn(8, 4), method="explicit_adams")
    with pytest.raises(ValueError, match="unknown ODE method"):
        sm.sample_ode_from_base(torch.randn(8, 4), method="rk45")
    # outside the compiled envelope (an activation the kernels do not implement): said once, and still GPU only
    from flowfusion_amd.fused import FusedEnvelopeWarning
    mish = D.ScoreModel(D.MLP(4, 0, 8, [64], activation=torch.nn.Mish()), D.VPSDE()).eval()
    with pytest.warns(FusedEnvelopeWarning), pytest.raises(RuntimeError, match="GPU"):
        mish.sample_ode_from_base(torch.randn(8, 4), method="euler")
    assert not mish._fusable() and sm._fusable()
    wide = F.ODEFlow(3, [4096, 64])
    with pytest.warns(FusedEnvelopeWarning), pytest.raises(RuntimeError, match="GPU"):
        wide.sample(torch.randn(4, 3), method="rk4", options={"step_size": 0.1})
    mish.precision = "bf16x3"                                    # an explicit arithmetic never switches silently
    with pytest.raises(NotImplementedError):
        mish.sample_ode_from_base(torch.randn(8, 4), method="euler")
    f = F.ODEFlow(3, [64, 64])
    with pytest.raises(RuntimeError, match="GPU"):
        f.sample(torch.randn(4, 3))                              # adaptive dopri5 default, as in the reference
    with pytest.raises(RuntimeError, match="GPU"):
        f.sample(torch.randn(4, 3), method="rk4", options={"step_size": 0.1})
    # the estimator solves: GPU only as well (adaptive default and fixed grid), and the flows' probe arguments are checked
    # before anything is launched
    hpp = D.ScoreModel(D.MLP(4, 0, 8, [64, 64]), D.VPSDE(), no_sigma=True, hutchpp=True).eval()
    with pytest.raises(RuntimeError, match="GPU"):
        hpp.log_prob(torch.randn(8, 4))
    with pytest.raises(RuntimeError, match="GPU"):
        hpp.log_prob(torch.randn(8, 4), method="rk4", options={"step_size": 0.25})
    with pytest.raises(ValueError, match="'torch' or 'philox'"):
        f.log_prob(torch.randn(4, 3), hutchinson=True, probe="sobol")
    with pytest.raises(ValueError, match="hutchinson=True"):
        f.log_prob(torch.randn(4, 3), probe="philox")
    with pytest.raises(ValueError, match="seed="):
        f.log_prob(torch.randn(4, 3), hutchinson=True, seed=3)


# ---- solvers -------------------------------------------------------------------------------------
def test_plan_ode_rows():
    eps = 1e-3
    plan = solvers.plan_ode(torch.tensor([1.0, eps]), "rk4", {"step_size": (1 - eps) / 10})
    g = O.grid_from_step_size(-torch.tensor([1.0, eps]), (1 - eps) / 10)
    assert plan.sign == -1.0 and plan.n_steps == g.numel() - 1
    assert plan.t_eval.numel() == 4 * plan.n_steps
    torch.testing.assert_close(plan.t_eval[0::4], -g[:-1], rtol=0, atol=0)     # first stage at t0
    torch.testing.assert_close(plan.t_eval[3::4], -g[1:], rtol=0, atol=0)      # last stage exactly at t1
    assert plan.flags.view(-1, 4)[:, 3].eq(solvers.FLAG_STEP_END).all() and plan.flags.view(-1, 4)[:, :3].eq(0).all()
    one = solvers.plan_ode(torch.tensor([0.0, 1.0]), "euler", None)             # no step_size: one step
    assert one.n_steps == 1 and float(one.cout[0, 0]) == 1.0
    with pytest.raises(NotImplementedError):
        solvers.plan_ode(torch.tensor([0.0, 1.0]), "dopri5", None)
    with pytest.raises(ValueError):
        solvers.plan_ode(torch.tensor([0.0, 1.0]), "nope", None)


def test_grid_constructor_and_perturb_options(built_library):
    """torchdiffeq's other fixed-grid options (FixedGridODESolver): ``grid_constructor(func, y0, t)`` in place of
    ``step_size`` (on a decreasing span torchdiffeq's `_check_inputs` wraps it as ``-gc(func, y0, -t)``: the user's
    function sees and returns REAL, decreasing times) and ``perturb=True`` (first stage one ulp
    after t0, a stage taken at t1 one ulp before it) -- rows checked word by word, then the kernel's semantics on the
    CPU against the oracle's restatement of the same options."""
    seen = {}

    def chebyshev_like(func, y0, t):
        seen["t"] = t.clone()
        u = torch.linspace(0, 1, 9, dtype=t.dtype) ** 2
        g = t[0] + (t[-1] - t[0]) * u
        g[-1] = t[-1]
        return g

    span = torch.tensor([1.0, 1e-3])
    plan = solvers.plan_ode(span, "rk4", {"grid_constructor": chebyshev_like, "perturb": True})
    assert seen["t"].tolist() == [1.0, 0.0010000000474974513]              # the real, decreasing span
    g = -chebyshev_like(None, None, span)                                  # solver time
    assert plan.sign == -1.0 and plan.n_steps == 8
    te = -plan.t_eval.view(8, 4)                                           # back to solver time
    assert torch.equal(te[:, 0], torch.nextafter(g[:-1], g[:-1] + 1)) and torch.equal(te[:, 3], torch.nextafter(g[1:], g[1:] - 1))
    assert torch.equal(te[:, 1], g[:-1] + (g[1:] - g[:-1]) * solvers._one_third)
    torch.testing.assert_close(plan.cout.view(8, 4, 8)[:, 3, 0], (g[1:] - g[:-1]) * 0.125, rtol=0, atol=0)
    plain = solvers.plan_ode(span, "midpoint", {"step_size": 0.25, "perturb": True})
    assert torch.equal(-plain.t_eval.view(-1, 2)[:, 0], torch.nextafter(torch.tensor([-1.0, -0.75, -0.5, -0.25]), torch.tensor(0.0)))
    with pytest.raises(ValueError, match="mutually exclusive"):
        solvers.plan_ode(span, "euler", {"step_size": 0.1, "grid_constructor": chebyshev_like})
    with pytest.raises(AssertionError):
        solvers.plan_ode(span, "euler", {"grid_constructor": lambda f, y, t: t * 0.5})
    with pytest.raises(ValueError, match="interpolation"):
        solvers.plan_ode(span, "euler", {"interp": "nope"})
    # an asymmetric grid written in REAL time on a decreasing span (what a torchdiffeq user writes): accepted as is,
    # the stage times of the plan are the real times, y0 is handed through
    got = {}

    def explicit(func, y0, t):
        got["y0"] = y0
        return torch.tensor([1.0, 0.5, 0.1, float(t[-1])])

    y0 = torch.zeros(3, 2)
    pe = solvers.plan_ode(span, "euler", {"grid_constructor": explicit}, y0=y0)
    assert got["y0"] is y0 and pe.sign == -1.0 and pe.n_steps == 3
    assert pe.t_eval.tolist() == [1.0, 0.5, 0.10000000149011612]
    torch.testing.assert_close(pe.cout[:, 0], torch.tensor([0.5, 0.4, 0.1 - 1e-3]), rtol=1e-6, atol=0)
    with pytest.raises(ValueError, match="monotonic"):
        solvers.plan_ode(span, "euler", {"grid_constructor": lambda f, y, t: torch.tensor([1.0, 0.2, 0.5, float(t[-1])])})
    inc = solvers.plan_ode(torch.tensor([1e-3, 1.0]), "euler", {"grid_constructor": lambda f, y, t: torch.tensor([float(t[0]), 0.3, 1.0])})
    assert inc.sign == 1.0 and inc.t_eval.tolist() == [0.0010000000474974513, 0.30000001192092896]

    torch.manual_seed(2)
    sm = D.ScoreModel(D.MLP(5, 2, 8, [64, 64]), D.VESDE(), no_sigma=False).eval()
    so = O.ScoreOracle(O.mlp_params_from_state_dict(dict(sm.state_dict()), "model."), O.VE(dtype=torch.float64), no_sigma=False,
                       dtype=torch.float64)
    base, cond = torch.randn(12, 5), torch.randn(12, 2)
    eps = float(sm.sde.epsilon)
    for method in ("euler", "heun3", "rk4"):
        opts = {"grid_constructor": chebyshev_like, "perturb": True}
        table = sm._ode_table(torch.tensor([1.0, eps]), method, opts, MODE_STATE)
        z = base * sm.sde.sigma_max
        got, _ = _emulate_score(sm, z, table, MODE_STATE, cond)
        ref = so.sample_ode_from_base(base.double(), cond.double(), method, opts)
        assert max_rel(got, ref, floor=ref.abs().max().item()) < 3e-5, method


def test_euler_maruyama_times():
    ts, dt = solvers.plan_euler_maruyama(1.0, torch.tensor(1e-3), 100)
    assert ts.numel() == 100 and ts[0] == 1.0
    assert abs(float(dt) + (1 - 1e-3) / 100) < 1e-9


# ---- state_dict compatibility + pointwise members ---------------------------------------------------
@pytest.mark.parametrize("name", golden_names("score_"))
def test_reference_state_dict_loads_and_pointwise_matches(name):
    meta, a = load_golden(name)
    sm = score_model(meta, a)                                     # strict load of the reference keys
    cond = a.get("cond")
    sm.conditional = cond
    for i in range(3):
        t = a[f"t{i}"]
        with torch.no_grad():
            got = sm.ode_drift(t, a["x"], conditional=cond)
        assert max_rel(got, a[f"xdot_{i}"], floor=a[f"xdot_{i}"].abs().max().item()) < 1e-5
        sm.prob, sm.hutch, sm.e = True, True, a["e"]
        _, div = sm.forward(t.clone(), (a["x"].clone(), torch.zeros(a["x"].shape[0], 1)))
        assert max_rel(div.detach(), a[f"div_hutch_{i}"], floor=a[f"div_hutch_{i}"].abs().max().item()) < 3e-5
        sm.hutch = False
        _, div = sm.forward(t.clone(), (a["x"].clone(), torch.zeros(a["x"].shape[0], 1)))
        assert max_rel(div.detach(), a[f"div_exact_{i}"], floor=a[f"div_exact_{i}"].abs().max().item()) < 3e-5
        sm.prob = False


@pytest.mark.parametrize("name", ["flow_3d", "flow_16d_ragged", "cflow_4d_c2", "cflow_8d_c5"])
def test_flow_state_dict_and_dynamics(name):
    meta, a = load_golden(name)
    f = flow_model(meta, a)
    data_keys = {"x", "cond", "t0", "t1", "t2"} | {k for k in a if k.startswith(("v_", "div_"))}
    assert set(f.state_dict().keys()) == set(a) - data_keys         # exactly the reference's keys
    cond = a.get("cond")
    for j in range(3):
        t = a[f"t{j}"]
        with torch.no_grad():
            v = f.dynamics(t, (a["x"], cond))[0] if cond is not None else f.dynamics(t, (a["x"],))
        torch.testing.assert_close(v, a[f"v_{j}"], rtol=1e-5, atol=1e-5)
        states = (a["x"].clone(), cond, torch.zeros(9, 1)) if cond is not None else (a["x"].clone(), torch.zeros(9, 1))
        div = f.dynamics_with_jacobian(t.clone(), states)[-1]
        torch.testing.assert_close(div.detach(), a[f"div_{j}"], rtol=3e-5, atol=3e-5)


def test_sde_members_match_golden():
    _, a = load_golden("sde_schedules")
    t, x = a["t"], a["x"]
    for key, sde in {"vp": D.VPSDE(), "ve": D.VESDE(), "subvp": D.SUBVPSDE(),
                     "vp_b": D.VPSDE(0.2, 12.0, 1.0, 1e-2), "ve_b": D.VESDE(0.05, 25.0, 1.0, 1e-4)}.items():
        torch.testing.assert_close(sde.sigma(t), a[f"{key}_sigma"], rtol=0, atol=0)
        torch.testing.assert_close(sde.diffusion(t, x), a[f"{key}_diffusion"], rtol=0, atol=0)
        torch.testing.assert_close(sde.drift(t, x), a[f"{key}_drift"], rtol=0, atol=0)
        m, s = sde.marginal_prob_scalars(t)
        torch.testing.assert_close(m, a[f"{key}_mean_scalar"], rtol=0, atol=0)
        torch.testing.assert_close(s, a[f"{key}_std_scalar"], rtol=0, atol=0)
        torch.testing.assert_close(sde.prior(x.shape).log_prob(x), a[f"{key}_prior_logprob"], rtol=1e-6, atol=1e-6)


# ---- packer + tables, end to end through the emulator -------------------------------------------------
def _emulate_score(sm, x, table, mode, cond=None, probe=None, noise=None):
    net = sm._net()
    plan = _native.plan_words(net.plan(mode))
    wpack = net.wpack("cpu", mode)
    return E.emulate(plan, wpack, table, x, cond=cond, probe=probe, noise=noise, mode=mode)


@pytest.mark.parametrize("name", golden_names("hybrid_score_"))
def test_emulated_kernel_matches_hybrid_golden(name, built_library):
    """packed weights + evaluation table, run through the kernel's semantics on the CPU, reproduce
    the reference-RHS x restated-stepper vectors (sampling, Hutchinson and exact log-density)."""
    meta, a = load_golden(name)
    sm = score_model(meta, a)
    cond = a.get("cond")
    eps = float(sm.sde.epsilon)
    for run in meta["runs"]:
        m, opts = run["method"], {"step_size": run["step_size"]}
        z = a["base"] * sm.sde.sigma_max if hasattr(sm.sde, "sigma_max") else a["base"]
        table = sm._ode_table(torch.tensor([1.0, eps]), m, opts, MODE_STATE)
        x0, _ = _emulate_score(sm, z, table, MODE_STATE, cond)
        exp = a[f"sample_{m}"]
        assert max_rel(x0, exp, floor=exp.abs().max().item()) < 3e-5, (name, m)
        xd = a[f"x_data_{m}"]
        tab_h = sm._ode_table(torch.tensor([eps, 1.0]), m, opts, MODE_HUTCH)
        xT, dlp = _emulate_score(sm, xd, tab_h, MODE_HUTCH, cond, probe=a[f"e_{m}"])
        lp = dlp[:, None] + sm.sde.prior(xT.shape).log_prob(xT.float()).double().sum(1, keepdim=True)
        assert max_rel(lp, a[f"lp_hutch_{m}"], floor=1.0) < 1e-4, (name, m)
        assert max_rel(xT, a[f"xT_{m}"], floor=a[f"xT_{m}"].abs().max().item()) < 3e-5
        if meta["D"] + 1 <= 32:
            tab_e = sm._ode_table(torch.tensor([eps, 1.0]), m, opts, MODE_EXACT)
            xT, dlp = _emulate_score(sm, xd, tab_e, MODE_EXACT, cond)
            lp = dlp[:, None] + sm.sde.prior(xT.shape).log_prob(xT.float()).double().sum(1, keepdim=True)
            assert max_rel(lp, a[f"lp_exact_{m}"], floor=1.0) < 1e-4, (name, m)


@pytest.mark.parametrize("name", golden_names("hybrid_flow") + golden_names("hybrid_cflow"))
def test_emulated_kernel_matches_hybrid_flow_golden(name, built_library):
    meta, a = load_golden(name)
    f = flow_model(meta, a)
    net = f._net()
    cond = a.get("cond")
    cn = None if cond is None else f._norm_cond(cond)
    for run in meta["runs"]:
        m, opts = run["method"], {"step_size": run["step_size"]}
        table = f._table(torch.tensor([1.0, 0.0]), m, opts, MODE_STATE)
        plan = _native.plan_words(net.plan(MODE_STATE))
        x0, _ = E.emulate(plan, net.wpack("cpu", MODE_STATE), table, a["xT"], cond=cn, mode=MODE_STATE,
                          out_scale=f.target_scale, out_shift=f.target_shift)
        torch.testing.assert_close(x0.float(), a[f"sample_{m}"], rtol=3e-5, atol=3e-5)
        table = f._table(torch.tensor([0.0, 1.0]), m, opts, MODE_EXACT)
        plan = _native.plan_words(net.plan(MODE_EXACT))
        xn = (a[f"x_data_{m}"] - f.target_shift) / f.target_scale
        xT, logj = E.emulate(plan, net.wpack("cpu", MODE_EXACT), table, xn, cond=cn, mode=MODE_EXACT)
        lp = torch.sum(-0.5 * xT ** 2 - 0.5 * torch.log(f.twopi.double()), dim=1) + logj - torch.log(f.target_scale.double()).sum()
        torch.testing.assert_close(lp.float(), a[f"logprob_{m}"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", [n for n in golden_names("sde_") if n != "sde_schedules"])
def test_emulated_euler_maruyama_matches_reference(name, built_library):
    """Table built for sample_sde + the reference's captured RNG draws reproduce the reference's
    own sample_sde output (diffusion.py:510-563)."""
    meta, a = load_golden(name)
    sm = score_model(meta, a)
    steps = meta["steps"]
    ts, dt = solvers.plan_euler_maruyama(torch.as_tensor(sm.sde.T, dtype=torch.float32), sm.sde.epsilon, steps)
    n = ts.numel()
    assert n == steps
    aa, bb, c1, g = sm._schedule(ts, "sde")
    cout = torch.zeros(n, 8)
    cout[:, 0] = dt
    flags = torch.full((n,), solvers.FLAG_STEP_END | solvers.FLAG_NOISE, dtype=torch.int32)
    flags[-1] = solvers.FLAG_STEP_END
    plan = solvers.EvalPlan(ts, 1.0, torch.zeros(n, dtype=torch.int32), flags, torch.zeros(n, 8), cout, n)
    table = solvers.build_table(plan, aa, bb, c1, sm._net().width(MODE_STATE), gn=g * (-dt) ** 0.5,
                                noise_idx=torch.arange(n))
    x, _ = _emulate_score(sm, a["x_prior"], table, MODE_STATE, a.get("cond"), noise=a["noise"])
    assert max_rel(x, a["out"], floor=a["out"].abs().max().item()) < 3e-5


def test_emulated_tile16_packing_matches_oracle(built_library):
    """The 16x16x4 tiling (wide / high-dimensional networks): packed weights decoded by the emulator
    reproduce the oracle for a 64-dim flow with 512-wide (and ragged) hidden layers."""
    torch.manual_seed(5)
    f = F.ConditionalODEFlow(target_dimension=40, conditional_dimension=5, hidden_units=[512, 200],
                             target_shift=torch.randn(40), target_scale=torch.rand(40) + 0.5,
                             conditional_shift=torch.randn(5), conditional_scale=torch.rand(5) + 0.5).eval()
    net = f._net()
    plan = _native.plan_words(net.plan(MODE_STATE))
    assert plan[7] == 16 and plan[3] == 512
    fo = flow_oracle({k: v.detach().clone() for k, v in f.state_dict().items()}, torch.float64)
    xT, cond = torch.randn(6, 40), torch.randn(6, 5)
    opts = {"step_size": 0.25}
    table = f._table(torch.tensor([1.0, 0.0]), "rk4", opts, MODE_STATE)
    x0, _ = E.emulate(plan, net.wpack("cpu", MODE_STATE), table, xT, cond=f._norm_cond(cond), mode=MODE_STATE,
                      out_scale=f.target_scale, out_shift=f.target_shift)
    torch.testing.assert_close(x0, fo.sample(xT.double(), cond.double(), "rk4", opts), rtol=1e-5, atol=1e-5)
    # Hutchinson mode on the same tiling
    e = torch.sign(torch.randn(6, 40))
    table = f._table(torch.tensor([0.0, 1.0]), "euler", opts, MODE_HUTCH)
    planh = _native.plan_words(net.plan(MODE_HUTCH))
    xn = (xT - f.target_shift) / f.target_scale
    xTT, logj = E.emulate(planh, net.wpack("cpu", MODE_HUTCH), table, xn, cond=f._norm_cond(cond), probe=e, mode=MODE_HUTCH)
    assert torch.isfinite(logj).all() and xTT.shape == xn.shape


ACTIVATIONS = [torch.nn.Tanh(), torch.nn.Sigmoid(), torch.nn.ReLU(), torch.nn.LeakyReLU(0.2), torch.nn.ELU(0.7),
               torch.nn.Softplus(), torch.nn.Softplus(beta=2.0, threshold=3.0), torch.nn.GELU(),
               torch.nn.GELU(approximate="tanh")]


KINKED = (torch.nn.ReLU, torch.nn.LeakyReLU, torch.nn.ELU)


@pytest.mark.parametrize("act", ACTIVATIONS, ids=lambda a: repr(a))
def test_activation_plumbing_matches_oracle(act, built_library):
    """`activation=` of the reference constructors: the module maps to an FF_ACT_* plan, and the kernel
    semantics with that activation (emulator) reproduce the oracle run with the same module."""
    torch.manual_seed(11)
    sm = D.ScoreModel(D.MLP(5, 2, 8, [48, 64], activation=act), D.VESDE(), no_sigma=False).eval()
    net = sm._net()
    plan = _native.plan_words(net.plan(MODE_EXACT))
    # the plan carries THIS activation; the kernel either has it compiled in (name suffix _a<FF_ACT code>, FF_BUILD_FULL)
    # or chooses it at run time from the plan (suffix _a9: one instantiation per width and mode since round 3)
    assert plan[8] == net.act[0] != _native.ACT_SILU
    assert built_library.ff_kernel_name(plan[6]).endswith((b"_a%d" % net.act[0], b"_a9"))
    so = O.ScoreOracle(O.mlp_params_from_state_dict(sm.state_dict()), O.VE(dtype=torch.float64), no_sigma=False,
                       dtype=torch.float64, activation=act)
    x0, cond = torch.randn(7, 5), torch.randn(7, 2)
    opts = {"step_size": 0.1}
    eps = float(sm.sde.epsilon)
    table = sm._ode_table(torch.tensor([eps, 1.0]), "rk4", opts, MODE_EXACT)
    xT, dl = _emulate_score(sm, x0, table, MODE_EXACT, cond)
    want_x, want_dl = so.solve_odes_forward(x0.double(), cond.double(), "rk4", opts, divergence="exact")
    torch.testing.assert_close(xT, want_x, rtol=2e-5, atol=2e-5)          # fp32 time table vs float64 oracle
    if not isinstance(act, KINKED):     # a slope that jumps at 0 makes the divergence discontinuous in the state
        torch.testing.assert_close(dl.view(-1), want_dl.view(-1), rtol=2e-5, atol=2e-5)
    # the flows take the activation as a class
    f = F.ODEFlow(3, [32, 32], activation=type(act)).eval()
    from flowfusion_amd.fused import activation_spec
    assert f._net().act == activation_spec(type(act)())


# ---- adaptive dopri5 driver (host logic) against the oracle's independent restatement -------------------
def _cpu_launcher(net, mode, cond=None, probe=None):
    plan = _native.plan_words(net.plan(mode))
    wpack = net.wpack("cpu", mode)
    return lambda y, k1, kl1, lp0, etab, n_aux, first, count: E.emulate_step(
        plan, wpack, etab, y, cond, probe, k1, kl1, lp0, mode, n_aux, first, count)


@pytest.mark.parametrize("case", ["ve_sample", "ve_hutch", "vp_exact_cond"])
def test_adaptive_dopri5_driver_matches_oracle(case, built_library):
    from flowfusion_amd import adaptive
    torch.manual_seed(3)
    if case == "ve_sample":
        meta = dict(D=6, C=0, E=8, units=[64, 64], sde="VESDE", sde_kw={}, no_sigma=True)
    elif case == "ve_hutch":
        meta = dict(D=3, C=0, E=8, units=[64, 64], sde="VESDE", sde_kw={}, no_sigma=False)
    else:
        meta = dict(D=4, C=2, E=8, units=[64], sde="VPSDE", sde_kw={}, no_sigma=False)
    m = D.MLP(meta["D"], meta["C"], meta["E"], meta["units"])
    sm = D.ScoreModel(m, getattr(D, meta["sde"])(), no_sigma=meta["no_sigma"]).eval()
    arrays = {k: v.detach().clone() for k, v in sm.state_dict().items()}
    so = score_oracle(meta, arrays, torch.float64)
    net = sm._net()
    B = 7
    x = torch.randn(B, meta["D"])
    cond = torch.randn(B, meta["C"]) if meta["C"] else None
    eps = float(sm.sde.epsilon)
    # Two adaptive solves agree to about the solver's global error, not to rounding: a last-bit difference in one
    # error norm can flip an accept/reject and move every later step.  So both run at a tolerance well below the
    # bar they are compared at (float64 oracle; the emulated kernel is float64 too, its tables fp32).
    rtol = atol = 1e-7
    TOL = 5e-5
    sched = lambda tr: sm._schedule(tr, "ode")[:3]
    if case == "ve_sample":
        x = x * sm.sde.sigma_max
        step = net.make_step(sched, -1.0, MODE_STATE, "cpu", launcher=_cpu_launcher(net, MODE_STATE))
        solver = adaptive.Dopri5(step, False, rtol, atol, None)
        t = torch.tensor([1.0, eps]).double()
        y, _ = solver.integrate(float(-t[0]), float(-t[1]), x, None)
        ref = so.sample_ode_from_base((x / sm.sde.sigma_max).double(), None, "dopri5", None, atol, rtol)
        assert solver.n_accepted >= 3
        assert max_rel(y, ref, floor=ref.abs().max().item()) < TOL
    else:
        mode = MODE_HUTCH if case == "ve_hutch" else MODE_EXACT
        e = torch.sign(torch.randn(B, meta["D"])) if mode == MODE_HUTCH else None
        step = net.make_step(sched, 1.0, mode, "cpu", cond=cond, probe=e, launcher=_cpu_launcher(net, mode, cond, e))
        # min_step below any step the controller asks for here: steps at or under min_step are accepted whatever
        # their error, and two implementations then agree only to that uncontrolled error
        solver = adaptive.Dopri5(step, True, rtol, atol, {"min_step": 1e-9})
        y, lp = solver.integrate(eps if False else float(torch.tensor(eps, dtype=torch.float32)), 1.0, x, torch.zeros(B))
        xT, dlp = so.solve_odes_forward(x.double(), None if cond is None else cond.double(), "dopri5", {"min_step": 1e-9},
                                        "hutch" if mode == MODE_HUTCH else "exact", None if e is None else e.double(),
                                        atol, rtol)
        assert solver.n_accepted >= 3
        assert max_rel(y, xT, floor=xT.abs().max().item()) < TOL
        assert max_rel(lp[:, None], dlp, floor=1.0) < TOL


@pytest.mark.parametrize("method", ["bosh3", "fehlberg2", "adaptive_heun"])
def test_other_embedded_pairs_match_oracle(method, built_library):
    """torchdiffeq's other adaptive solvers with at most 7 stages run on the same driver (one launch per attempted step,
    different tableau).  Emulated kernel vs the oracle's independent restatement, tolerances well below the bar;
    also: the pair converges to the dopri5 answer (a wrong coefficient would not)."""
    from flowfusion_amd import adaptive
    torch.manual_seed(5)
    meta = dict(D=3, C=0, E=8, units=[64, 64], sde="VESDE", sde_kw={}, no_sigma=False)
    sm = D.ScoreModel(D.MLP(3, 0, 8, [64, 64]), D.VESDE(), no_sigma=False).eval()
    so = score_oracle(meta, {k: v.detach().clone() for k, v in sm.state_dict().items()}, torch.float64)
    net = sm._net()
    B = 5
    x = torch.randn(B, 3) * 0.5
    e = torch.sign(torch.randn(B, 3))
    eps = float(torch.tensor(float(sm.sde.epsilon), dtype=torch.float32))
    sched = lambda tr: sm._schedule(tr, "ode")[:3]
    rtol = atol = 1e-6 if method != "bosh3" else 1e-7
    step = net.make_step(sched, 1.0, MODE_HUTCH, "cpu", probe=e, launcher=_cpu_launcher(net, MODE_HUTCH, None, e))
    solver = adaptive.Dopri5(step, True, rtol, atol, {"min_step": 1e-9}, method=method)
    y, lp = solver.integrate(eps, 1.0, x, torch.zeros(B))
    xT, dlp = so.solve_odes_forward(x.double(), None, method, {"min_step": 1e-9}, "hutch", e.double(), atol, rtol)
    assert solver.n_accepted >= 5
    assert max_rel(y, xT, floor=xT.abs().max().item()) < 2e-4 and max_rel(lp[:, None], dlp, floor=1.0) < 2e-4
    x5, d5 = so.solve_odes_forward(x.double(), None, "dopri5", {"min_step": 1e-9}, "hutch", e.double(), 1e-9, 1e-9)
    assert max_rel(y, x5, floor=x5.abs().max().item()) < 1e-3 and max_rel(lp[:, None], d5, floor=1.0) < 1e-3


def test_dopri8_stage_by_stage_matches_oracle(built_library):
    """torchdiffeq's dopri8 (13 stages + FSAL: more than the 7 stage slots / 8 row coefficients of the fused kernels) on the
    stage-by-stage driver, adaptive.HostSteppedPair: every stage one emulated single-row launch, every combination a pass
    over the state.  Sampling direction (reversed time) and Hutchinson log-density against the oracle's restatement of the
    same pair, and convergence to the dopri5 answer; make_solver picks the driver by method."""
    from flowfusion_amd import adaptive
    torch.manual_seed(5)
    meta = dict(D=3, C=0, E=8, units=[64, 64], sde="VESDE", sde_kw={}, no_sigma=False)
    sm = D.ScoreModel(D.MLP(3, 0, 8, [64, 64]), D.VESDE(), no_sigma=False).eval()
    so = score_oracle(meta, {k: v.detach().clone() for k, v in sm.state_dict().items()}, torch.float64)
    net = sm._net()
    B = 5
    x = torch.randn(B, 3) * 0.5
    e = torch.sign(torch.randn(B, 3))
    eps = float(torch.tensor(float(sm.sde.epsilon), dtype=torch.float32))
    sched = lambda tr: sm._schedule(tr, "ode")[:3]
    rtol = atol = 1e-7
    step = net.make_step(sched, 1.0, MODE_HUTCH, "cpu", probe=e, launcher=_cpu_launcher(net, MODE_HUTCH, None, e))
    solver = adaptive.make_solver(step, True, rtol, atol, {"min_step": 1e-9}, method="dopri8")
    assert isinstance(solver, adaptive.HostSteppedPair) and solver.tab.stages == 14
    assert type(adaptive.make_solver(step, True, rtol, atol, None, method="dopri5")) is adaptive.Dopri5
    y, lp = solver.integrate(eps, 1.0, x, torch.zeros(B))
    xT, dlp = so.solve_odes_forward(x.double(), None, "dopri8", {"min_step": 1e-9}, "hutch", e.double(), atol, rtol)
    # (counts are not compared: at 1e-7 the fp32 rounding of the table rows is the size of the error estimate itself, and
    # the float64 oracle takes fewer attempts over the same span)
    assert solver.n_accepted >= 3 and O.last_adaptive_stats["accepted"] >= 3
    assert max_rel(y, xT, floor=xT.abs().max().item()) < 5e-5 and max_rel(lp[:, None], dlp, floor=1.0) < 5e-5
    x5, d5 = so.solve_odes_forward(x.double(), None, "dopri5", {"min_step": 1e-9}, "hutch", e.double(), 1e-9, 1e-9)
    assert max_rel(y, x5, floor=x5.abs().max().item()) < 1e-4 and max_rel(lp[:, None], d5, floor=1.0) < 1e-4
    # sampling: decreasing span, state only
    z = torch.randn(B, 3) * sm.sde.sigma_max
    step = net.make_step(sched, -1.0, MODE_STATE, "cpu", launcher=_cpu_launcher(net, MODE_STATE))
    solver = adaptive.make_solver(step, False, rtol, atol, None, method="dopri8")
    ys, _ = solver.integrate(-1.0, -eps, z, None)
    ref = so.sample_ode_from_base((z / sm.sde.sigma_max).double(), None, "dopri8", None, atol, rtol)
    assert max_rel(ys, ref, floor=ref.abs().max().item()) < 5e-5
    with pytest.raises(NotImplementedError, match="HostSteppedPair"):
        adaptive.Dopri5(step, False, rtol, atol, None, method="dopri8")


def test_adaptive_nan_error_estimate_raises_like_torchdiffeq(built_library):
    """VP schedules are undefined for t < 0; an overshooting last step makes the error estimate NaN,
    upon which torchdiffeq asserts ('underflow in dt nan').  The driver must not spin."""
    from flowfusion_amd import adaptive
    torch.manual_seed(3)
    sm = D.ScoreModel(D.MLP(6, 0, 8, [64, 64]), D.VPSDE(), no_sigma=True).eval()
    net = sm._net()
    step = net.make_step(lambda tr: sm._schedule(tr, "ode")[:3], -1.0, MODE_STATE, "cpu",
                         launcher=_cpu_launcher(net, MODE_STATE))
    solver = adaptive.Dopri5(step, False, 1e-5, 1e-5, None)
    with pytest.raises(RuntimeError, match="underflow in dt"):
        solver.integrate(-1.0, -0.001, torch.randn(7, 6), None)


# ---- in-kernel noise: the published generator and the header's mapping -----------------------------------
def test_philox_known_answers_and_normals():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors) pin the restatement the GPU tests compare the
    kernel with; the Box-Muller mapping gives standard normals."""
    import numpy as np
    from tests._philox import normals, philox4x32_10
    kat = [([0, 0, 0, 0], (0, 0), [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, (0xffffffff, 0xffffffff), [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0xa4093822, 0x299f31d0),
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        got = philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert [int(v) for v in got] == want
    z = normals(seed=2024, sample_offset=0, batch=100000, dim=6, noise_indices=[0, 1, 2])
    assert z.shape == (3, 100000, 6) and np.isfinite(z).all()
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3 and abs((z ** 4).mean() - 3) < 5e-2
    # keyed by the global row index: a shard sees exactly its rows of the whole-batch stream
    part = normals(2024, 4000, 1000, 6, [1])
    assert np.array_equal(part[0], z[1, 4000:5000])
    # distinct streams per noise index, dimension and seed
    assert abs(np.corrcoef(z[0, :, 0], z[1, :, 0])[0, 1]) < 0.02 and abs(np.corrcoef(z[0, :, 0], z[0, :, 1])[0, 1]) < 0.02
    assert not np.array_equal(normals(2025, 0, 10, 6, [0]), z[:1, :10])


def test_emulated_euler_maruyama_with_philox_noise(built_library):
    """Host plumbing of noise="philox": the table of a whole Euler-Maruyama run (one launch) through the kernel
    semantics with the restated generator equals the oracle's loop fed with the same normals."""
    import numpy as np
    from tests._philox import normals
    torch.manual_seed(3)
    sm = D.ScoreModel(D.MLP(6, 2, 8, [64, 64]), D.VESDE(), no_sigma=False).eval()
    so = O.ScoreOracle(O.mlp_params_from_state_dict(sm.state_dict()), O.VE(dtype=torch.float64), no_sigma=False,
                       dtype=torch.float64)
    B, steps, seed, off = 9, 12, 77, 1000
    x = torch.randn(B, 6) * 10
    cond = torch.randn(B, 2)
    captured = {}

    real = sm._net
    net = real()

    class _Net:
        def __getattr__(self, k):
            return getattr(net, k)

        def integrate(self, x, table, mode, cond=None, noise=None, rng=None, **kw):
            captured["rng"] = rng
            plan = _native.plan_words(net.plan(mode))
            y, _ = E.emulate(plan, net.wpack("cpu", mode), table, x, cond=cond, mode=mode, rng=rng)
            return y.float(), None, torch.zeros(1, dtype=torch.int32)

    sm._net = lambda: _Net()
    try:
        got = sm._sample_sde_from(x, None, cond, steps, rng=(seed, off))
    finally:
        sm._net = real
    assert captured["rng"] == (seed, off, 0)
    z = torch.from_numpy(normals(seed, off, B, 6, list(range(steps)))).double()
    want = so.sample_sde(x.double(), [z[i] for i in range(steps)], cond.double(), steps=steps)
    torch.testing.assert_close(got.double(), want, rtol=2e-5, atol=2e-5 * float(want.abs().max()))


# ---- PopulationModel wrappers: reference state_dict layout and sample_sde (fixtures pop*.npz) -----------------
def _population_model(meta, a, device="cpu"):
    m = D.MLP(n_dimensions=meta["D"], n_conditionals=meta["C"], embedding_dimensions=meta["E"], units=meta["units"])
    if meta["C"]:
        pm = D.PopulationModelDiffusionConditional(model=m, sde=D.VESDE())
    else:
        pm = D.PopulationModelDiffusion(model=m, sde=D.VESDE())
    data = {"x_prior", "noise", "out", "cond"}
    pm.load_state_dict({k: v for k, v in a.items() if k not in data}, strict=True)      # exactly the reference's keys
    return pm.to(device).eval()


@pytest.mark.parametrize("name", ["pop_4d", "popcond_4d_c2"])
def test_population_wrappers_load_reference_state_and_sample(name, built_library):
    meta, a = load_golden(name)
    pm = _population_model(meta, a)
    sm = pm.score_model
    cond = a.get("cond")
    cn = None if cond is None else (cond - pm.conditional_shift) / pm.conditional_scale      # diffusion.py:1808-1809
    T, eps = torch.as_tensor(sm.sde.T, dtype=torch.float32), sm.sde.epsilon.detach()
    ts, dt = solvers.plan_euler_maruyama(T, eps, 100)                 # the wrappers always take 100 steps (:1608, :1810)
    aa, bb, c1, g = sm._schedule(ts, "sde")
    n = ts.numel()
    flags = torch.full((n,), solvers.FLAG_STEP_END | solvers.FLAG_NOISE, dtype=torch.int32)
    flags[-1] = solvers.FLAG_STEP_END
    cout = torch.zeros(n, 8)
    cout[:, 0] = dt
    plan = solvers.EvalPlan(t_eval=ts, sign=1.0, slot=torch.zeros(n, dtype=torch.int32), flags=flags,
                            cin=torch.zeros(n, 8), cout=cout, n_steps=n)
    table = solvers.build_table(plan, aa, bb, c1, sm._net().width(MODE_STATE), gn=g * (-dt) ** 0.5,
                                noise_idx=torch.arange(n))
    x, _ = _emulate_score(sm, a["x_prior"], table, MODE_STATE, cn, noise=a["noise"])
    got = x * pm.scale.double() + pm.shift.double()                   # :1606-1609
    assert max_rel(got, a["out"], floor=a["out"].abs().max().item()) < 3e-5


def test_replaced_layer_invalidates_the_fused_view(built_library):
    """A layer swapped after the first solve (layer surgery, re-initialisation) must be repacked: the kernel-side
    view is rebuilt when ANY Linear module changed identity, not only the first.  Also: the state-only and the
    divergence kernels of one shape share one packed buffer (the layout does not depend on the instantiation)."""
    sm = D.ScoreModel(D.MLP(4, 0, 8, [64, 64, 64]), D.VPSDE(), no_sigma=True).eval()
    net = sm._net()
    w0 = net.wpack("cpu", MODE_STATE).clone()
    assert sm._net() is net
    assert net.wpack("cpu", MODE_HUTCH) is net.wpack("cpu", MODE_STATE)
    torch.manual_seed(123)
    sm.model.NN[1] = torch.nn.Linear(64, 64)
    net2 = sm._net()
    assert net2 is not net and net2.linears[1] is sm.model.NN[1]
    w1 = net2.wpack("cpu", MODE_STATE)
    assert w1.shape == w0.shape and not torch.equal(w0, w1)
    # in-place updates keep the view and refresh the pack
    with torch.no_grad():
        sm.model.NN[2].weight.mul_(2.0)
    assert sm._net() is net2 and not torch.equal(net2.wpack("cpu", MODE_STATE), w1)
    # a changed activation rebuilds it as well
    sm.model.activation = torch.nn.Tanh()
    assert sm._net() is not net2
    f = F.ODEFlow(3, [64, 64])
    nf = f._net()
    f.layers[2] = torch.nn.Linear(64, 64)
    assert f._net() is not nf


def test_other_dtypes_are_refused_not_rounded(built_library):
    """The reference follows the dtype of its parameters and inputs (a `.double()` model solves in float64 there); the
    kernels compute in fp32 only, so a float64 model or input is refused with a TypeError instead of being rounded."""
    from flowfusion_amd import flow as Fm
    sm = D.ScoreModel(D.MLP(4, 0, 8, [64]), D.VPSDE()).eval()
    with pytest.raises(TypeError, match="float32"):
        sm.sample_ode_from_base(torch.randn(8, 4, dtype=torch.float64), method="rk4", options={"step_size": 0.1})
    with pytest.raises(TypeError, match="float32"):
        sm.double().log_prob(torch.randn(8, 4), method="rk4", options={"step_size": 0.1})
    with pytest.raises(TypeError, match="float32"):
        sm.sample_sde((8, 4), steps=3)
    f = Fm.ODEFlow(3, [64, 64]).eval().double()
    with pytest.raises(TypeError, match="float32"):
        f.sample(torch.randn(4, 3))


def test_launch_kind_states_the_launchers_rule(built_library, monkeypatch):
    """ff_mlp_launch_kind (no GPU needed): the launcher's choice between the one-wavefront kernel, its cooperative twin and
    both, as a query -- what the GPU tests assert instead of a wall-clock ratio.  Config 2's network: 16 samples per tile,
    two wavefronts per SIMD, 2,048 tiles in flight."""
    from flowfusion_amd import _native
    monkeypatch.delenv("FF_COOP", raising=False)
    monkeypatch.delenv("FF_TAIL_SPLIT", raising=False)
    p0 = _native.make_plan(16, 0, [256] * 4, 0)
    kinds = {b: _native.launch_kind(p0, b, 0) for b in (1, 2048, 12288, 16384, 32768, 32768 + 300, 1 << 20)}
    T, W, B = _native.LAUNCH_TWIN, _native.LAUNCH_ONE_WAVE, _native.LAUNCH_ONE_WAVE_AND_TWIN
    assert kinds == {1: T, 2048: T, 12288: T, 16384: W, 32768: W, 32768 + 300: B, 1 << 20: W}
    assert _native.launch_kind(p0, 32768 + 300, 0, jac_out=True) == W              # a Jacobian output is never split
    monkeypatch.setenv("FF_TAIL_SPLIT", "0")
    assert _native.launch_kind(p0, 32768 + 300, 0) == W
    monkeypatch.setenv("FF_COOP", "0")
    assert _native.launch_kind(p0, 2048, 0) == W
    monkeypatch.setenv("FF_COOP", "1")
    assert _native.launch_kind(p0, 1 << 20, 0) == T
    monkeypatch.delenv("FF_COOP")
    # tangent columns change the samples per tile (Hutchinson: 8; exact trace with 15 tangents: 1)
    p1 = _native.make_plan(16, 0, [256] * 4, 1)
    assert _native.launch_kind(p1, 8 * 700, 1) == T and _native.launch_kind(p1, 8 * 4096, 1) == W
    p2 = _native.make_plan(16, 0, [256] * 4, 2)
    assert _native.launch_kind(p2, 700, 2, tangent_count=15) == T
    # the wide catch-all is cooperative at every size
    pw = _native.make_plan(100, 40, [700, 700], 0)
    assert _native.launch_kind(pw, 1 << 18, 0) == T
    with pytest.raises(RuntimeError, match="FF_ERR_BADARG"):
        _native.launch_kind(p0, -1, 0)


def test_adaptive_options_step_t_jump_t_norm(built_library):
    """torchdiffeq's remaining adaptive options on the host controller (the reference hands `options` through untouched,
    diffusion.py:631-639, 744-752): `step_t` -- accepted steps END on the given times; `jump_t` -- the same and one more
    right-hand side per jump; `norm` -- a user callable on the tuple state (x, delta_logp [B, 1]).  Kernel-semantics
    emulator against the oracle's restatement (decreasing span: the time lists are negated with it), attempt for attempt;
    and against a property neither restatement can fake: with the max-norm as `norm` every component's scaled error
    stays below 1 in every accepted step."""
    from flowfusion_amd import adaptive
    torch.manual_seed(8)
    meta = dict(D=3, C=0, E=8, units=[64, 64], sde="VESDE", sde_kw={}, no_sigma=False)
    sm = D.ScoreModel(D.MLP(3, 0, 8, [64, 64]), D.VESDE(), no_sigma=False).eval()
    so = score_oracle(meta, {k: v.detach().clone() for k, v in sm.state_dict().items()}, torch.float64)
    net = sm._net()
    B = 6
    eps = float(torch.tensor(float(sm.sde.epsilon), dtype=torch.float32))
    sched = lambda tr: sm._schedule(tr, "ode")[:3]
    z = torch.randn(B, 3) * float(sm.sde.sigma_max)

    # sampling, decreasing span 1 -> eps: step_t / jump_t are REAL times, negated inside
    def sample(options):
        step = net.make_step(sched, -1.0, MODE_STATE, "cpu", launcher=_cpu_launcher(net, MODE_STATE))
        solver = adaptive.make_solver(step, False, 1e-5, 1e-5, dict(options), method="dopri5", sign=-1.0)
        seen = []
        orig = solver._attempt
        solver._attempt = lambda ta, dt, tb, *a: (seen.append((ta, tb)), orig(ta, dt, tb, *a))[1]
        y, _ = solver.integrate(-1.0, -eps, z.clone(), None)
        return y, solver, seen
    plain, s0, _ = sample({})
    pts = [0.75, 0.4, 0.9, 1.5]                                      # (1.5 lies outside the span: ignored)
    y, s1, seen = sample({"step_t": torch.tensor(pts)})
    ends = {round(-tb, 12) for _, tb in seen}
    f32 = lambda v: round(float(torch.tensor(v, dtype=torch.float32)), 12)      # (the option came as an fp32 tensor)
    assert {f32(0.9), 0.75, f32(0.4)} <= ends and s1.n_accepted >= s0.n_accepted
    ref = so.sample_ode_from_base((z / sm.sde.sigma_max).double(), None, "dopri5", {"step_t": torch.tensor(pts)}, 1e-5, 1e-5)
    assert O.last_adaptive_stats["attempts"] == s1.n_attempts and O.last_adaptive_stats["accepted"] == s1.n_accepted
    assert max_rel(y, ref, floor=ref.abs().max().item()) < 5e-5
    assert max_rel(y, plain, floor=plain.abs().max().item()) < 1e-3      # the same solution, other steps
    yj, sj, seenj = sample({"jump_t": [0.6], "step_t": [0.3]})
    assert {0.6, 0.3} <= {round(-tb, 12) for _, tb in seenj}
    refj = so.sample_ode_from_base((z / sm.sde.sigma_max).double(), None, "dopri5", {"jump_t": [0.6], "step_t": [0.3]}, 1e-5, 1e-5)
    assert (O.last_adaptive_stats["attempts"], O.last_adaptive_stats["accepted"]) == (sj.n_attempts, sj.n_accepted)
    assert max_rel(yj, refj, floor=refj.abs().max().item()) < 5e-5
    with pytest.raises(ValueError, match="repeated elements"):
        sample({"jump_t": [0.6], "step_t": [0.6]})

    # log-density, increasing span, user norm = max norm over the tuple (x, delta_logp [B, 1])
    x0 = torch.randn(B, 3) * 0.5
    e = torch.sign(torch.randn(B, 3))
    shapes = []

    def max_norm(state):
        shapes.append(tuple(tuple(c.shape) for c in state))
        return max(c.abs().max() for c in state)
    step = net.make_step(sched, 1.0, MODE_HUTCH, "cpu", probe=e, launcher=_cpu_launcher(net, MODE_HUTCH, None, e))
    solver = adaptive.make_solver(step, True, 1e-4, 1e-4, {"norm": max_norm, "min_step": 1e-9}, method="dopri5")
    ratios = []
    orig_norms = solver._norms
    solver._norms = lambda terms, check=None: (lambda r: (ratios.append(r[0]) if check is not None else None, r)[1])(orig_norms(terms, check))
    y, lp = solver.integrate(eps, 1.0, x0.clone(), torch.zeros(B))
    assert set(shapes) == {((B, 3), (B, 1))}
    xT, dlp = so.solve_odes_forward(x0.double(), None, "dopri5", {"norm": max_norm, "min_step": 1e-9}, "hutch", e.double(), 1e-4, 1e-4)
    assert (O.last_adaptive_stats["attempts"], O.last_adaptive_stats["accepted"]) == (solver.n_attempts, solver.n_accepted)
    assert max_rel(y, xT, floor=xT.abs().max().item()) < 2e-4 and max_rel(lp[:, None], dlp, floor=1.0) < 2e-4
    # under the max norm an accepted step has EVERY scaled error component <= 1; the default RMS norm accepts steps whose
    # worst component is above it -- so the max norm must have taken at least as many steps
    step2 = net.make_step(sched, 1.0, MODE_HUTCH, "cpu", probe=e, launcher=_cpu_launcher(net, MODE_HUTCH, None, e))
    rms = adaptive.make_solver(step2, True, 1e-4, 1e-4, {"min_step": 1e-9}, method="dopri5")
    rms.integrate(eps, 1.0, x0.clone(), torch.zeros(B))
    assert solver.n_accepted >= rms.n_accepted and len(ratios) == solver.n_attempts
    with pytest.raises(TypeError, match="callable"):
        adaptive.make_solver(step2, True, 1e-4, 1e-4, {"norm": "rms"}, method="dopri5")
