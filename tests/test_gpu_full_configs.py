"""GPU tier: BASELINE.json's configurations at their FULL step counts (200-step flows, 1000-step
Euler-Maruyama, the 50-step demo) and the PopulationModel wrappers, through the product classes (hence the C
ABI), against the CPU oracle at sizes it finishes in seconds and through size-independent properties at the
per-GPU batch of the configuration.

Tolerances: fixed grids are held to 2e-5 against the float64 oracle like everywhere else in the tier (states
relative to the largest reference magnitude, log-densities relative with floor 1); two adaptive solves to 2e-4.
Long horizons matter here: 1200 sequential network evaluations (config 4) and 1000 noisy steps (config 5) are
where a rounding-order difference would have had room to grow.
"""
import pytest
import torch

from tests._util import flow_oracle, max_rel, score_oracle
from tests.test_gpu_parity import (ADAPT_TOL, DEV, LOGP_TOL, STATE_TOL, _logp_err, _seeded_score_model, _state_err)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(built_library):
    assert torch.cuda.is_available(), "the gpu tier needs a GPU"


# ---- config 4: 64-dim flow matching, MLP 5x512, 200 fixed steps ------------------------------------------------
def _config4_flow(seed):
    from flowfusion_amd import flow as Fm
    torch.manual_seed(seed)
    f = Fm.ODEFlow(target_dimension=64, hidden_units=[512] * 5, target_shift=torch.randn(64),
                   target_scale=torch.rand(64) + 0.5).eval()
    sd = {k: v.detach().clone() for k, v in f.state_dict().items()}
    return f.to(DEV), flow_oracle(sd), flow_oracle(sd, torch.float64)


@pytest.mark.parametrize("method", ["dopri5_fixed", "rk4"])
def test_config4_200_steps_against_oracle(method):
    """flow.py:259-306 with 200 fixed steps: 1200 (Dormand-Prince) / 800 (3/8-rule) evaluations of the 5x512 net."""
    f, fo32, fo64 = _config4_flow(131)
    B = 64
    torch.manual_seed(7)
    xT = torch.randn(B, 64)
    opts = {"step_size": 1.0 / 200}
    got = f.sample(xT.to(DEV), method=method, options=opts)
    ref64 = fo64.sample(xT.double(), None, method, opts).float()
    e_gpu = _state_err(got, ref64)
    e_cpu = _state_err(fo32.sample(xT, None, method, opts), ref64)
    assert e_gpu < STATE_TOL, (method, e_gpu, e_cpu)
    # and the log-density at 200 steps on a handful of points (exact divergence: 64 unit tangents, 5 launches)
    if method == "rk4":
        x = got[:6].cpu()
        lp = f.log_prob(x.to(DEV), method=method, options=opts)
        ref = fo64.log_prob(x.double(), None, method, opts).float()
        assert _logp_err(lp, ref) < LOGP_TOL


def test_config4_full_size_200_steps_properties():
    """2^22 samples over 8 GPUs = 2^19 per GPU, 200-step fixed Dormand-Prince: determinism, batch-shape
    invariance (bitwise), oracle on a subsample, ODE reversibility."""
    f, fo32, _ = _config4_flow(132)
    B = 1 << 19
    g = torch.Generator(device=DEV).manual_seed(5)
    xT = torch.randn(B, 64, device=DEV, generator=g)
    opts = {"step_size": 1.0 / 200}
    x = f.sample(xT, method="dopri5_fixed", options=opts)
    assert torch.isfinite(x).all()
    assert torch.equal(x, f.sample(xT, method="dopri5_fixed", options=opts))
    for sl in (slice(0, 100), slice(B // 2 - 3, B // 2 + 70), slice(B - 77, B)):
        assert torch.equal(f.sample(xT[sl].contiguous(), method="dopri5_fixed", options=opts), x[sl])
    idx = torch.arange(0, B, B // 16)
    assert _state_err(x[idx], fo32.sample(xT[idx].cpu(), None, "dopri5_fixed", opts)) < STATE_TOL
    # integrate the samples forward again (t: 0 -> 1): the base points come back
    y = (x - f.target_shift) / f.target_scale
    back, _ = f._solve(y, torch.tensor([0.0, 1.0]), "dopri5_fixed", opts, 0, 1e-5, 1e-5)
    assert ((back - xT).abs().max() / xT.abs().max()).item() < 1e-4


# ---- config 5: conditional 32-dim VE, MLP 4x256, 1000-step Euler-Maruyama ---------------------------------------
def test_config5_em_1000_steps_against_oracle():
    """diffusion.py:510-563 at 1000 steps with an injected random stream, B = 32, vs the oracle's loop."""
    sm, so32, so64 = _seeded_score_model(32, 8, [256] * 4, "VESDE", False, 141)
    B, steps = 32, 1000
    torch.manual_seed(3)
    prior = torch.randn(B, 32) * float(sm.sde.sigma_max)
    cond = torch.randn(B, 8)
    noise = torch.randn(steps, B, 32)
    it = iter(noise)
    got = sm._sample_sde_from(prior.to(DEV), lambda like: next(it).to(DEV), cond.to(DEV), steps=steps)
    ref64 = so64.sample_sde(prior.double(), [n.double() for n in noise], cond.double(), steps=steps).float()
    e_gpu = _state_err(got, ref64)
    e_cpu = _state_err(so32.sample_sde(prior, list(noise), cond, steps=steps), ref64)
    assert e_gpu < STATE_TOL, (e_gpu, e_cpu)


def test_config5_full_size_1000_steps_properties():
    """2^17 samples, all 1000 steps, the reference's random stream (torch draws, 16 noise buffers' worth, drawn on
    a side stream): bitwise re-run under the same seed; a subset re-run alone with the same draws gives the same
    bits; and the in-kernel stream (one launch for the 1000 steps) is shard invariant."""
    sm, _, _ = _seeded_score_model(32, 8, [256] * 4, "VESDE", False, 142)
    Bs, steps = 1 << 17, 1000
    g = torch.Generator(device=DEV).manual_seed(9)
    cond = torch.randn(Bs, 8, device=DEV, generator=g)
    torch.manual_seed(17)
    a = sm.sample_sde((Bs, 32), conditional=cond, steps=steps)
    torch.manual_seed(17)
    b = sm.sample_sde((Bs, 32), conditional=cond, steps=steps)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    # first 200 rows alone: draw the same full-size slabs in the same order, hand over their first rows
    torch.manual_seed(17)
    prior = sm.sde.prior([32]).sample([Bs]).to(DEV)
    small = sm._sample_sde_from(prior[:200].contiguous(), lambda like: torch.randn_like(prior)[:200].contiguous(),
                                cond[:200].contiguous(), steps=steps)
    assert torch.equal(small, a[:200])
    # counter-based stream: the whole run is one launch; cutting the batch changes nothing
    p = sm._sample_sde_from(prior, None, cond, steps, rng=(77, 0))
    cut = 50000
    q = sm._sample_sde_from(prior[cut:].contiguous(), None, cond[cut:].contiguous(), steps, rng=(77, cut))
    assert torch.equal(q, p[cut:]) and torch.isfinite(p).all()
    # the two streams sample the same distribution
    assert abs(float(p.mean() - a.mean())) < 0.05 * float(a.std()) and abs(float(p.std() / a.std()) - 1) < 0.03


def test_em_nan_stop_and_progress_callback(capsys):
    """diffusion.py:560-563: at the first step whose state holds a NaN the reference prints, stops and returns
    that step's mean.  One poisoned noise value makes one sample's state NaN at a known step; the oracle's loop
    (same break rule) is the expectation.  Also: the progress callback sees every chunk."""
    sm, so32, _ = _seeded_score_model(4, 0, [64, 64], "VPSDE", True, 143)
    B, steps = 300, 40
    torch.manual_seed(1)
    prior = torch.randn(B, 4)
    noise = torch.randn(steps, B, 4)
    seen = []
    it = iter(noise)
    ok = sm._sample_sde_from(prior.to(DEV), lambda like: next(it).to(DEV), None, steps,
                             progress=lambda done, total: seen.append((done, total)), progress_every=8)
    assert seen == [(8, 40), (16, 40), (24, 40), (32, 40), (40, 40)]
    assert _state_err(ok, so32.sample_sde(prior, list(noise), None, steps=steps)) < STATE_TOL
    # poison one sample's noise at step 13: its state is NaN after that step, the loop must stop there
    bad = noise.clone()
    bad[13, 7, 2] = float("nan")
    it = iter(bad)
    got = sm._sample_sde_from(prior.to(DEV), lambda like: next(it).to(DEV), None, steps)
    assert "NaN were produced" in capsys.readouterr().out
    ref = so32.sample_sde(prior, list(bad), None, steps=steps)        # breaks at step 13, returns its x_mean
    assert torch.isfinite(ref).all()                                  # the mean of that step is still clean
    assert _state_err(got, ref) < STATE_TOL


# ---- config 1: the 2-D demo --------------------------------------------------------------------------------------
@pytest.mark.parametrize("sde_name,units,no_sigma", [("VPSDE", [64] * 3, False), ("VPSDE", [64] * 3, True),
                                                      ("VESDE", [128] * 3, False)])
def test_config1_2d_50_step_euler(sde_name, units, no_sigma):
    """BASELINE configs[0] as worded (2-D VP-SDE, 3x64, 50-step Euler) and as the notebook has it (2-D VE,
    3x128, demo_diffusion.ipynb:189-193): the 50-step Euler probability-flow sampler and the 50-step
    Euler-Maruyama sampler, 50,000 points like the notebook (:388), oracle on a subsample."""
    sm, so32, so64 = _seeded_score_model(2, 0, units, sde_name, no_sigma, 151)
    B = 50000
    g = torch.Generator(device=DEV).manual_seed(2)
    base = torch.randn(B, 2, device=DEV, generator=g)
    opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / 50}
    x, _ = sm.sample_ode_from_base(base, method="euler", options=opts)
    idx = torch.arange(0, B, 97)
    ref64 = so64.sample_ode_from_base(base[idx].cpu().double(), None, "euler", opts).float()
    assert _state_err(x[idx], ref64) < STATE_TOL
    sm.hutch = True
    torch.manual_seed(4)
    lp = sm.log_prob(x[idx], method="euler", options=opts)
    ref = so64.log_prob(x[idx].cpu().double(), None, "euler", opts, "hutch", sm.e.cpu().double()).float()
    assert _logp_err(lp, ref) < LOGP_TOL
    # sample_sde, 50 steps, with the draws the public method makes on this device
    torch.manual_seed(6)
    s = sm.sample_sde((B, 2), steps=50)
    torch.manual_seed(6)
    prior = sm.sde.prior([2]).sample([B]).to(DEV)
    draws = [torch.randn_like(prior)[idx].cpu() for _ in range(50)]
    ref = so32.sample_sde(prior[idx].cpu(), draws, None, steps=50)
    assert _state_err(s[idx], ref) < STATE_TOL


# ---- PopulationModel wrappers --------------------------------------------------------------------------------------
def _population(conditional, sde_name, no_sigma, hutch, method, options, seed):
    from flowfusion_amd import diffusion as Dm
    from oracle import flowfusion_oracle as O
    torch.manual_seed(seed)
    D, C = 5, (3 if conditional else 0)
    mlp = Dm.MLP(D, C, 8, [96, 64])
    sde = getattr(Dm, sde_name)()
    shift, scale = torch.randn(D), torch.rand(D) + 0.5
    if conditional:
        cshift, cscale = torch.randn(C), torch.rand(C) + 0.5
        pm = Dm.PopulationModelDiffusionConditional(model=mlp, sde=sde, shift=shift, scale=scale, conditional_shift=cshift,
                                                    conditional_scale=cscale, no_sigma=no_sigma, method=method, options=options)
    else:
        cshift = cscale = None
        pm = Dm.PopulationModelDiffusion(model=mlp, sde=sde, shift=shift, scale=scale, method=method, no_sigma=no_sigma,
                                         hutchinson=hutch, options=options)
    pm = pm.eval()
    meta = dict(D=D, C=C, E=8, units=[96, 64], sde=sde_name, sde_kw={}, no_sigma=no_sigma)
    arrays = {k: v.detach().clone() for k, v in pm.score_model.state_dict().items()}
    mk = lambda dt: O.PopulationOracle(score_oracle(meta, arrays, dt), shift, scale, cshift, cscale, method, options)
    return pm.to(DEV), mk(torch.float32), mk(torch.float64)


@pytest.mark.parametrize("conditional", [False, True])
@pytest.mark.parametrize("sde_name,no_sigma", [("VPSDE", True), ("VESDE", False)])
def test_population_wrappers_forward_and_log_prob_against_oracle(conditional, sde_name, no_sigma):
    """PopulationModelDiffusion[Conditional].forward / .log_prob (diffusion.py:1556-1640, 1754-1848) against the
    oracle's restatement: output affine (in the kernel epilogue on a fixed grid), input affine (kernel prologue),
    conditional normalisation, `self.method` honoured by forward and ignored by log_prob."""
    opts = {"step_size": 0.02}
    pm, po32, po64 = _population(conditional, sde_name, no_sigma, False, "rk4", opts, 161)
    B = 120
    torch.manual_seed(5)
    base = torch.randn(B, 5)
    cond = torch.randn(B, 3) * 1.5 + 0.3 if conditional else None
    args = () if cond is None else (cond.to(DEV),)
    out = pm(base.to(DEV), *args)
    ref = po64.forward(base.double(), None if cond is None else cond.double()).float()
    assert _state_err(out, ref) < STATE_TOL
    # the fold changes no bit: the epilogue is x * scale + shift in two roundings, like the torch expression
    raw, _ = pm.score_model.sample_ode_from_base(base.to(DEV), conditional=None if cond is None else pm._cond(cond.to(DEV)),
                                                 method="rk4", options=opts)
    assert torch.equal(out, raw * pm.scale + pm.shift)
    # log_prob: exact trace, adaptive dopri5 whatever self.method says; options = self.options travel along
    x = out[:40].cpu()
    c40 = None if cond is None else cond[:40]
    a40 = () if cond is None else (c40.to(DEV),)
    lp = pm.log_prob(x.to(DEV), *a40)
    assert lp.shape == (40, 1)
    assert pm.score_model.last_solver_stats["accepted"] >= 3          # it really was the adaptive solver
    ref = po32.log_prob(x, c40)
    assert _logp_err(lp, ref) < ADAPT_TOL
    # forward with the reference's default method (dopri5): affine applied around the adaptive loop.  (VE here; the
    # random-init VP network's reverse flow grows without bound and the adaptive solver gives up -- "underflow in dt
    # nan" -- in product and oracle alike, after the same number of steps: pinned by
    # test_gpu_device_adaptive.py::test_diverging_vp_reverse_flow_is_pinned_on_both_sides.)
    if sde_name != "VESDE":
        return
    pm.method, pm.options = "dopri5", None
    po32.method, po32.options = "dopri5", None
    out = pm(base.to(DEV), *args)
    assert _state_err(out, po32.forward(base, cond)) < ADAPT_TOL


@pytest.mark.parametrize("prec", ["bf16x3", "bf16x2"])
def test_population_wrappers_under_split_precision(prec):
    """The wrappers' precision= keyword reaches the inner ScoreModel: forward (fixed grid, affine in the epilogue), the default
    log_prob (adaptive dopri5 + exact trace, affine in front of the loop) and sample_sde on the split-precision kernels
    agree with the f32 kernels' results."""
    from flowfusion_amd import diffusion as Dm, _native
    torch.manual_seed(191)
    shift, scale = torch.randn(5), torch.rand(5) + 0.5
    cshift, cscale = torch.randn(3), torch.rand(3) + 0.5
    pm = Dm.PopulationModelDiffusionConditional(model=Dm.MLP(5, 3, 8, [96, 128]), sde=Dm.VESDE(), shift=shift, scale=scale,
                                                conditional_shift=cshift, conditional_scale=cscale, method="rk4",
                                                options={"step_size": 0.02}, precision=prec).eval().to(DEV)
    assert pm.score_model.precision == prec
    base, cond = torch.randn(90, 5, device=DEV), torch.randn(90, 3, device=DEV)
    out = pm(base, cond)
    assert _native.kernel_name(pm.score_model._net().plan(0)).startswith("mlp_ode_split")
    if prec == "bf16x3":          # three parts: state-only kernels (round 3); a log-density never switches arithmetic silently
        with pytest.raises(NotImplementedError, match="bf16x3"):
            pm.log_prob(out[:30], cond[:30])
        lp = None
    else:
        lp = pm.log_prob(out[:30], cond[:30])
    pm.score_model.precision = "f32"
    out32 = pm(base, cond)
    lp32 = pm.log_prob(out[:30], cond[:30])
    assert _state_err(out, out32.cpu()) < STATE_TOL
    # (two adaptive solves at rtol = atol = 1e-5 on right-hand sides that differ at 1e-6: similar, not equal, step sequences)
    assert lp is None or _logp_err(lp, lp32.cpu()) < ADAPT_TOL


def test_population_wrapper_hutchinson_log_prob_fixed_seed():
    pm, po32, _ = _population(False, "VPSDE", True, True, "dopri5", None, 171)
    torch.manual_seed(8)
    x = torch.randn(64, 5) * 0.7
    torch.manual_seed(21)
    lp = pm.log_prob(x.to(DEV), atol=1e-5, rtol=1e-5)
    e = pm.score_model.e.cpu()
    torch.manual_seed(21)
    assert torch.equal(e, torch.sign(torch.randn(64, 5)))             # CPU draw, diffusion.py:701
    assert _logp_err(lp, po32.log_prob(x, None, 1e-5, 1e-5, "hutch", e)) < ADAPT_TOL


def test_population_wrapper_with_float64_statistics():
    """shift / scale computed with numpy arrive as float64 buffers: the reference's `solve * scale + shift` then promotes the
    samples to float64 (diffusion.py:1575-1585), and so does the wrapper here (the affine leaves the kernel's epilogue for
    that case); the float32 solve underneath is the same.  A float64 input to the network is refused (the reference's
    Linear layers raise on it as well)."""
    from flowfusion_amd.diffusion import MLP, VESDE, PopulationModelDiffusion
    torch.manual_seed(3)
    net = MLP(5, 0, 8, [128, 128])
    shift, scale = torch.randn(5), torch.rand(5) + 0.5
    opts = {"step_size": 0.05}
    p32 = PopulationModelDiffusion(net, VESDE(), shift=shift, scale=scale, method="rk4", options=opts).eval().to(DEV)
    p64 = PopulationModelDiffusion(net, VESDE(), shift=shift.double(), scale=scale.double(), method="rk4", options=opts).eval().to(DEV)
    z = torch.randn(300, 5, device=DEV)
    a, b = p32(z), p64(z)
    assert a.dtype == torch.float32 and b.dtype == torch.float64
    assert (a.double() - b).abs().max().item() < 1e-5 * b.abs().max().item()
    with pytest.raises(TypeError, match="float32"):
        p64.log_prob(b)


def test_replaced_layer_is_repacked():
    """a layer swapped after the first solve must not keep integrating with the old weights."""
    sm, _, _ = _seeded_score_model(16, 0, [64, 64, 64], "VPSDE", True, 181)
    opts = {"step_size": 0.1}
    z = torch.randn(64, 16, device=DEV)
    a, _ = sm.sample_ode_from_base(z, method="rk4", options=opts)
    torch.manual_seed(0)
    sm.model.NN[2] = torch.nn.Linear(64, 64).to(DEV)
    b, _ = sm.sample_ode_from_base(z, method="rk4", options=opts)
    assert not torch.equal(a, b)
    import copy
    fresh = copy.deepcopy(sm)
    fresh._fused = None
    c, _ = fresh.sample_ode_from_base(z, method="rk4", options=opts)
    assert torch.equal(b, c)


@pytest.mark.parametrize("method", ["bosh3", "fehlberg2", "adaptive_heun"])
def test_other_adaptive_methods_run_natively(method):
    """`method=` is passed through to torchdiffeq by the reference (diffusion.py:631-639): its other embedded pairs
    with at most 7 stages run on the same one-launch-per-attempt driver; against the oracle's restatement."""
    sm, so32, _ = _seeded_score_model(4, 0, [128, 128], "VESDE", False, 191)
    torch.manual_seed(8)
    base = torch.randn(200, 4)
    x, _ = sm.sample_ode_from_base(base.to(DEV), method=method, atol=1e-5, rtol=1e-5)
    assert sm.last_solver_stats["accepted"] >= 5
    assert _state_err(x, so32.sample_ode_from_base(base, None, method, None, 1e-5, 1e-5)) < 5e-4
    x0 = torch.randn(48, 4) * 0.5
    sm.hutch = True
    torch.manual_seed(5)
    lp = sm.log_prob(x0.to(DEV), method=method, atol=1e-5, rtol=1e-5)
    ref = so32.log_prob(x0, None, method, {"min_step": 1e-6}, "hutch", sm.e.cpu(), 1e-5, 1e-5)
    assert _logp_err(lp, ref) < 5e-4


def test_dopri8_runs_stage_by_stage():
    """`method="dopri8"` reaches torchdiffeq through the reference's pass-through (diffusion.py:631-639, flow.py:371-382).
    13 stages + FSAL do not fit the fused kernels' 7 stage slots, so an attempted step is walked stage by stage (one fused
    single-row launch per stage, ff_stage_combine for the algebra: adaptive.HostSteppedPair) under the same step control:
    sampling, Hutchinson and exact-trace log-density, a flow, and any other `model=` module, against the oracle's
    restatement of the pair (tableau verified by its order conditions, tests/test_oracle_known_answers.py)."""
    from flowfusion_amd import flow as Fm
    from oracle import flowfusion_oracle as O
    sm, so32, _ = _seeded_score_model(4, 2, [128, 128], "VESDE", False, 191)
    torch.manual_seed(8)
    base, cond = torch.randn(200, 4), torch.randn(200, 2)
    x, _ = sm.sample_ode_from_base(base.to(DEV), conditional=cond.to(DEV), method="dopri8", atol=1e-5, rtol=1e-5)
    st = dict(sm.last_solver_stats)
    assert st["accepted"] >= 3 and "chunks" not in st                       # the host-stepped driver, not the device controller
    assert _state_err(x, so32.sample_ode_from_base(base, cond, "dopri8", None, 1e-5, 1e-5)) < ADAPT_TOL
    x5, _ = sm.sample_ode_from_base(base.to(DEV), conditional=cond.to(DEV), atol=1e-6, rtol=1e-6)      # dopri5
    assert _state_err(x, x5.cpu()) < ADAPT_TOL and sm.last_solver_stats["attempts"] > st["attempts"]    # higher order, fewer steps
    x0 = torch.randn(48, 4) * 0.5
    for hutch in (True, False):
        sm.hutch = hutch
        torch.manual_seed(5)
        lp = sm.log_prob(x0.to(DEV), conditional=cond[:48].to(DEV), method="dopri8", atol=1e-5, rtol=1e-5)
        ref = so32.log_prob(x0, cond[:48], "dopri8", {"min_step": 1e-6}, "hutch" if hutch else "exact", sm.e.cpu() if hutch else None,
                            1e-5, 1e-5)
        assert _logp_err(lp, ref) < ADAPT_TOL, hutch
    sm.hutch = False
    torch.manual_seed(31)
    f = Fm.ODEFlow(3, [64, 64]).eval()
    fo = O.FlowOracle(O.flow_params_from_state_dict({k: v.detach().clone() for k, v in f.state_dict().items()}))
    f = f.to(DEV)
    xT = torch.randn(100, 3)
    assert _state_err(f.sample(xT.to(DEV), method="dopri8", atol=1e-6, rtol=1e-6), fo.sample(xT, None, "dopri8", None, 1e-6, 1e-6)) < ADAPT_TOL
    lpf = f.log_prob(xT.to(DEV) * 0.7, method="dopri8")
    assert _logp_err(lpf, fo.log_prob(xT * 0.7, None, "dopri8", None)) < ADAPT_TOL


# ---- bench.py's N > 1 branch ---------------------------------------------------------------------------------------
def test_bench_four_ranks_rehearsal_on_one_gpu():
    """The driver launches bench.py for N > 1 as `python -m torch.distributed.run ... bench.py --gpus N`.  No second
    GPU exists here, so the multi-rank branch (process group, sharded solve, the all-gather inside the timed region,
    max-over-ranks timing, per-rank report, the sharded BASELINE configs[2] / [3] / [4] extras with their all-gathers, the
    rank-invariance checks, the adaptive path's whole-batch step control) is rehearsed with FOUR ranks on cuda:0 over gloo, in a fresh child process tree (started
    before this process hands anything to it; nothing is exec'ed after GPU init).  Four, not eight: a GPU box admits
    at most six processes on its card, and the N = 8 launch is the driver's; nothing in the branch depends on the world
    size beyond `shard_bounds`, which tests/test_distributed_gloo.py runs at world 8 on the CPU."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    W = 4
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(W), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(root / "bench.py"), "--gpus", str(W), "--backend", "gloo", "--single-device",
           "--steps", "1", "--warmup", "0", "--cpu-batch", "0", "--batch", "16384",
           "--c3-rows", "6100", "--c3-steps", "6", "--adaptive-rows", "3001",
           "--c4-rows", "4100", "--c4-steps", "8", "--c5-rows", "8200", "--c5-steps", "24"]       # ragged shards on purpose
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # rank 0 prints ONE line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == W and rec["scaling"] == "weak" and rec["config"]["global_batch"] == W * 16384
    assert rec["value"] > 0 and rec["steps"] == 1
    assert len(rec["per_rank"]["kernel_ms_avg"]) == W and all(v > 0 for v in rec["per_rank"]["kernel_ms_avg"])
    assert rec["rank_invariant"] is True and rec["host_threads_per_rank"] >= 1
    assert "cpu_baseline" not in rec
    extras = rec["extra_configs"]
    assert len(extras) == 4 and all(f"configs[{i + 2}]" in e["workload"] for i, e in enumerate(extras[:3]))
    ad = extras[3]                                                      # the adaptive path's exchange step
    assert "error" not in ad and ad["steps_equal_whole_batch_solve_on_all_ranks"] is True, ad
    assert len(ad["per_rank"]["attempts"]) == W and max(ad["per_rank"]["max_rel_diff_vs_whole_batch_solve"]) < 2e-5, ad
    for e, rows in zip(extras[:3], (6100, 4100, 8200)):
        assert e["rank_invariant"] is True and e["value"] > 0
        for k in ("wall_ms", "kernel_ms", "allgather_ms"):
            assert len(e["per_rank"][k]) == W and all(v >= 0 for v in e["per_rank"][k])
        assert e["rows_per_rank"] == -(-rows // W)                      # rank 0 holds the larger share of a ragged split


def test_rccl_single_rank(tmp_path):
    """RCCL itself, as far as one GPU allows: a one-rank "nccl" process group in a fresh child process runs every collective
    the multi-GPU code issues and the adaptive path's exchange hook (the all-reduce enqueued from the C driver's callback
    between the reduction launch and the controller launch) -- see tests/_rccl_worker.py."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, FF_RESULT_DIR=str(tmp_path), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(root / "tests" / "_rccl_worker.py")], capture_output=True, text=True, timeout=600,
                       cwd=str(root), env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    rec = json.loads((tmp_path / "rccl.json").read_text())
    assert rec and all(rec.values()), rec


def test_bench_multi_rank_branch_under_rccl_with_one_rank():
    """bench.py's N > 1 branch with backend "nccl" (RCCL), as far as one GPU allows: `--group-of-one` takes that branch with a
    one-rank process group, so its device-side collectives (the all-gather inside the timed region with HIP events around
    it, the float64 all-reduces / all-gathers of the per-rank report, the sharded extras, the guarded adaptive extra) have
    all run under RCCL before a node runs them."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "1", "--group-of-one", "--steps", "2", "--warmup", "1",
           "--cpu-batch", "0", "--batch", "16384", "--c3-rows", "6100", "--c3-steps", "6", "--adaptive-rows", "3001",
           "--c4-rows", "4100", "--c4-steps", "8", "--c5-rows", "8200", "--c5-steps", "24"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(root), env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["rank_invariant"] is True and rec["value"] > 0
    assert len(rec["per_rank"]["kernel_ms_avg"]) == 1 and rec["per_rank"]["allgather_ms_avg"][0] >= 0
    extras = rec["extra_configs"]
    assert len(extras) == 4 and all("error" not in e for e in extras), extras
    assert all(e["rank_invariant"] for e in extras[:3]) and extras[3]["steps_equal_whole_batch_solve_on_all_ranks"] is True


def test_bench_line_survives_a_collective_that_never_completes():
    """The N > 1 extras run behind a watchdog: here rank 1 stays away from the last one, rank 0 waits for it in a collective
    for ever -- after the watchdog's patience (8 s in this test) rank 0 must still print the ONE line, with the headline
    intact and an `error` entry in place of the extra (saying which section, on which rank) -- and then leave with exit code 3
    (bench.EXIT_EXTRAS_FAILED), so that the launcher's return code tells this run from a clean one."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
           "--steps", "1", "--warmup", "0", "--cpu-batch", "0", "--batch", "8192", "--c3-rows", "2100", "--c3-steps", "4",
           "--c4-rows", "2100", "--c4-steps", "4", "--c5-rows", "2100", "--c5-steps", "8", "--adaptive-rows", "2001",
           "--watchdog-seconds", "600", "8", "--test-desert-rank", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(root))
    import re
    assert r.returncode != 0, (r.stdout[-1500:], r.stderr[-3000:])          # torch.distributed.run reports the failed children
    assert re.search(r"exitcode\s*:\s*3\b", r.stderr), r.stderr[-3000:]       # ... and their code is the bench's own
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["rank_invariant"] is True
    extras = rec["extra_configs"]
    assert len(extras) == 4 and all("error" not in e for e in extras[:3])
    assert "no answer within" in extras[3]["error"]
    assert extras[3]["where"]["rank"] == 0 and extras[3]["where"]["watchdog_fired"] is True


def test_offsets_beyond_two_gib():
    """2^25 x 16 fp32 = 2 GiB per array: element offsets cross 2^31 bytes in the fused kernel, the counter-based fill and the
    streaming helpers.  Slices of the huge batch solved on their own return its rows bit for bit (first, middle, last)."""
    from flowfusion_amd import _native
    from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel
    torch.manual_seed(5)
    sm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True).eval().to(DEV)
    eps = float(sm.sde.epsilon)
    B = 1 << 25
    z = _native.normal_fill(B, 16, 7, 0, DEV)
    assert torch.equal(_native.normal_fill(100, 16, 7, B - 100, DEV), z[B - 100:])
    o = {"step_size": (1.0 - eps) / 2}
    x, _ = sm.sample_ode_from_base(z, method="euler", options=o)
    for lo in (0, B // 2 - 5, B // 2 + 12345, B - 1000):
        part, _ = sm.sample_ode_from_base(z[lo:lo + 1000].contiguous(), method="euler", options=o)
        assert torch.equal(part, x[lo:lo + 1000]), lo
    got = _native.scaled_rms([(x, None, z, x)], 1e-3, 1e-3, check=x)
    tail = slice(B - (1 << 20), B)
    ref_tail = ((x[tail].double() / (1e-3 + 1e-3 * torch.max(z[tail].abs(), x[tail].abs()).double())) ** 2).sum()
    ref_head = ((x[: B - (1 << 20)].double() / (1e-3 + 1e-3 * torch.max(z[: B - (1 << 20)].abs(), x[: B - (1 << 20)].abs()).double())) ** 2).sum()
    ref = float(((ref_tail + ref_head) / x.numel()).sqrt())
    assert abs(got[0] - ref) < 1e-5 * ref and got[1] == 0.0
    keep = (x[-1000:] + 0.5 * z[-1000:]).clone()
    _native.stage_combine(z, x, [z], [0.5], 1.0)                         # in place over 2 GiB: z <- x + 0.5 z
    assert torch.equal(z[-1000:], keep)
