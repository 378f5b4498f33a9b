"""GPU tier: the device-resident adaptive solver (csrc/ff_adaptive.hip, flowfusion_amd/device_adaptive.py) -- the
reference's DEFAULT solver at every call site (torchdiffeq dopri5: flowfusion/diffusion.py:572, 649, 762;
flowfusion/flow.py:299-303, 313) with the step control on the device.

* against the host controller (adaptive.Dopri5, FF_HOST_CONTROLLER=1): the SAME attempt / accept counts and the same
  answers to fp32 rounding of the table rows (device vs libm transcendentals), for every schedule kind, mode, tangent
  pass count, embedded pair and option the front ends pass through;
* against the CPU oracle's restatement of torchdiffeq (2e-4, like the other adaptive tests);
* torchdiffeq's assertions ("underflow in dt", "max_num_steps exceeded") raised with the host controller's words and
  counts -- including the diverging random-init VP reverse flow of round 2 (`r02a`), pinned on both sides;
* the two kernel-level guards that came with it: ff_ode_args.gate (a launch behind the end of a solve is a no-op) and
  FF_STATUS_BAD_SLOT (a row naming a stage slot the kernel does not keep is refused).
"""
import ctypes

import pytest
import torch

from tests.test_gpu_parity import ADAPT_TOL, DEV, _logp_err, _seeded_score_model, _state_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(built_library):
    assert torch.cuda.is_available(), "the gpu tier needs a GPU"


def _both(monkeypatch, fn):
    """fn() under the device controller and under the host controller: ((result, stats), (result, stats))."""
    monkeypatch.delenv("FF_HOST_CONTROLLER", raising=False)
    dev = fn()
    monkeypatch.setenv("FF_HOST_CONTROLLER", "1")
    host = fn()
    monkeypatch.delenv("FF_HOST_CONTROLLER", raising=False)
    return dev, host


def _same_counts(a, b):
    return (a["attempts"], a["accepted"]) == (b["attempts"], b["accepted"])


SCORE_CASES = {
    # name: (D, C, units, sde, no_sigma, B)
    "notebook_2d_ve_3x128": (2, 0, [128] * 3, "VESDE", False, 5000),
    "c2_16d_vp_4x256": (16, 0, [256] * 4, "VPSDE", True, 700),
    "cond_5d_c3_subvp_ragged": (5, 3, [64, 100], "SUBVPSDE", False, 333),
    "ve_no_sigma_8d_c2": (8, 2, [128, 128], "VESDE", True, 257),
}


@pytest.mark.parametrize("name", list(SCORE_CASES))
def test_score_models_device_controller_equals_host_controller(name, monkeypatch):
    """Default-argument calls (adaptive dopri5; sampling, Hutchinson and exact-trace log-density -- 16 dimensions take
    two unit-tangent passes per attempted step) under both controllers, and against the oracle."""
    D, C, units, sde_name, no_sigma, B = SCORE_CASES[name]
    sm, so32, _ = _seeded_score_model(D, C, units, sde_name, no_sigma, 901)
    torch.manual_seed(17)
    base = torch.randn(B, D)
    cond = torch.randn(B, C) if C else None
    cd = None if cond is None else cond.to(DEV)

    def sample():
        try:
            x, _ = sm.sample_ode_from_base(base.to(DEV), conditional=cd)
        except RuntimeError as e:          # a random-init VP reverse flow may diverge: torchdiffeq's assertion, on both sides
            return str(e), e.solver_stats
        return x, dict(sm.last_solver_stats)
    (xd, sd), (xh, sh) = _both(monkeypatch, sample)
    assert _same_counts(sd, sh), (sd, sh, xd if isinstance(xd, str) else "", xh if isinstance(xh, str) else "")
    assert "chunks" in sd and "chunks" not in sh                       # it really was the device loop / the host loop
    if isinstance(xd, str) or isinstance(xh, str):
        assert xd == xh and "VESDE" not in sde_name, (xd, xh)
        with pytest.raises(AssertionError, match="underflow in dt"):
            so32.sample_ode_from_base(base, cond, "dopri5", None)
    else:
        assert sd["accepted"] >= 3
        assert _state_err(xd, xh.cpu()) < 1e-5
        assert _state_err(xd, so32.sample_ode_from_base(base, cond, "dopri5", None)) < ADAPT_TOL
    n = min(B, 48)
    x0 = torch.randn(n, D) * 0.5
    c0 = None if cond is None else cond[:n]
    c0d = None if c0 is None else c0.to(DEV)
    for hutch in (True, False):
        sm.hutch = hutch

        def logp():
            torch.manual_seed(5)
            lp = sm.log_prob(x0.to(DEV), conditional=c0d)
            return lp, dict(sm.last_solver_stats)
        (ld, sd), (lh, sh) = _both(monkeypatch, logp)
        assert _same_counts(sd, sh), (name, hutch, sd, sh)
        assert _logp_err(ld, lh.cpu()) < 1e-5, (name, hutch)
        e = sm.e.cpu() if hutch else None
        ref = so32.log_prob(x0, c0, "dopri5", {"min_step": 1e-6}, "hutch" if hutch else "exact", e)
        assert _logp_err(ld, ref) < ADAPT_TOL, (name, hutch)
    sm.hutch = False


def test_flows_device_controller_equals_host_controller(monkeypatch):
    """ODEFlow / ConditionalODEFlow default calls: sample (dopri5 at torchdiffeq's default tolerances), log_prob with the
    exact trace (5 passes at 64 dimensions) and with a Hutchinson probe; the raw conditional rides in the solver state
    (norm-only component of the initial-step rule, flow.py:779-796)."""
    from flowfusion_amd import flow as Fm
    from oracle import flowfusion_oracle as O
    torch.manual_seed(31)
    f = Fm.ODEFlow(3, [64, 64]).eval()
    fo = O.FlowOracle(O.flow_params_from_state_dict({k: v.detach().clone() for k, v in f.state_dict().items()}))
    f = f.to(DEV)
    xT = torch.randn(400, 3)

    def sample():
        return f.sample(xT.to(DEV)), dict(f.last_solver_stats)
    (a, sa), (b, sb) = _both(monkeypatch, sample)
    assert _same_counts(sa, sb) and sa["accepted"] >= 3 and _state_err(a, b.cpu()) < 1e-5
    assert _state_err(a, fo.sample(xT, None, "dopri5", None)) < ADAPT_TOL
    x = torch.randn(60, 3) * 0.7

    def logp():
        return f.log_prob(x.to(DEV)), dict(f.last_solver_stats)
    (a, sa), (b, sb) = _both(monkeypatch, logp)
    assert _same_counts(sa, sb) and _logp_err(a, b.cpu()) < 1e-5
    assert _logp_err(a, fo.log_prob(x, None, "dopri5", None)) < ADAPT_TOL
    # conditional, with non-trivial affines: the raw conditional is a norm-only component
    torch.manual_seed(32)
    g = Fm.ConditionalODEFlow(4, 2, [96, 96], target_shift=torch.randn(4), target_scale=torch.rand(4) + 0.5,
                              conditional_shift=torch.randn(2) * 3, conditional_scale=torch.rand(2) + 0.5).eval()
    go = O.FlowOracle(O.flow_params_from_state_dict({k: v.detach().clone() for k, v in g.state_dict().items()}))
    g = g.to(DEV)
    xT, cond = torch.randn(130, 4), torch.randn(130, 2) * 40.0         # large raw values: they dominate d0

    def csample():
        return g.sample(xT.to(DEV), cond.to(DEV)), dict(g.last_solver_stats)
    (a, sa), (b, sb) = _both(monkeypatch, csample)
    assert _same_counts(sa, sb) and _state_err(a, b.cpu()) < 1e-5
    assert _state_err(a, go.sample(xT, cond, "dopri5", None)) < ADAPT_TOL

    def clogp():
        torch.manual_seed(3)
        return g.log_prob(xT[:40].to(DEV), cond[:40].to(DEV), hutchinson=True), dict(g.last_solver_stats)
    (a, sa), (b, sb) = _both(monkeypatch, clogp)
    assert _same_counts(sa, sb) and _logp_err(a, b.cpu()) < 1e-5
    # 64 dimensions, exact trace: five unit-tangent passes per attempted step, summed on the device
    torch.manual_seed(33)
    h = Fm.ODEFlow(64, [128, 128]).eval().to(DEV)
    xw = torch.randn(24, 64) * 0.5

    def wlogp():
        return h.log_prob(xw.to(DEV), atol=1e-4, rtol=1e-4), dict(h.last_solver_stats)
    (a, sa), (b, sb) = _both(monkeypatch, wlogp)
    assert _same_counts(sa, sb) and _logp_err(a, b.cpu()) < 1e-5


@pytest.mark.parametrize("method", ["bosh3", "fehlberg2", "adaptive_heun"])
def test_other_embedded_pairs_on_the_device_controller(method, monkeypatch):
    sm, so32, _ = _seeded_score_model(4, 0, [128, 128], "VESDE", False, 191)
    torch.manual_seed(8)
    base = torch.randn(200, 4)

    def run():
        x, _ = sm.sample_ode_from_base(base.to(DEV), method=method, atol=1e-5, rtol=1e-5)
        return x, dict(sm.last_solver_stats)
    (a, sa), (b, sb) = _both(monkeypatch, run)
    assert _same_counts(sa, sb) and sa["accepted"] >= 5 and _state_err(a, b.cpu()) < 1e-5
    assert sa["chunks"] >= 2 or sa["attempts"] <= 16                    # long solves take several chunks
    assert _state_err(a, so32.sample_ode_from_base(base, None, method, None, 1e-5, 1e-5)) < 5e-4


def test_options_and_assertions_match_the_host_controller(monkeypatch):
    """min_step / max_step / first_step / max_num_steps through `options=`; torchdiffeq's assertions come back with the
    host controller's words and counts."""
    sm, so32, _ = _seeded_score_model(6, 0, [128, 128], "VESDE", False, 77)
    torch.manual_seed(2)
    base = torch.randn(150, 6)
    for opts in ({"first_step": 0.004}, {"max_step": 0.05}, {"min_step": 0.01}, {"first_step": 0.5, "max_step": 0.1}):
        def run():
            x, _ = sm.sample_ode_from_base(base.to(DEV), options=dict(opts))
            return x, dict(sm.last_solver_stats)
        (a, sa), (b, sb) = _both(monkeypatch, run)
        assert _same_counts(sa, sb), (opts, sa, sb)
        assert _state_err(a, b.cpu()) < 1e-5, opts
        assert _state_err(a, so32.sample_ode_from_base(base, None, "dopri5", dict(opts))) < ADAPT_TOL, opts

    def capped():
        try:
            sm.sample_ode_from_base(base.to(DEV), options={"max_num_steps": 4})
        except RuntimeError as e:
            return str(e), e.solver_stats
        return None, None
    (ma, sa), (mb, sb) = _both(monkeypatch, capped)
    assert ma == mb == "max_num_steps exceeded (4>=4)" and _same_counts(sa, sb) and sa["attempts"] == 4
    with pytest.raises(NotImplementedError, match="multistep"):
        sm.sample_ode_from_base(base.to(DEV), method="implicit_adams")
    # step_t / jump_t / norm need the host (steps ending on given times, a Python callable): such solves take the host
    # controller by themselves, and agree with the oracle run with the same options
    for opts in ({"step_t": torch.tensor([0.5, 0.25])}, {"jump_t": [0.7], "step_t": [0.1]},
                 {"norm": lambda state: max(c.abs().max() for c in state)}):
        x, _ = sm.sample_ode_from_base(base.to(DEV), options=dict(opts))
        assert "chunks" not in sm.last_solver_stats and sm.last_solver_stats["accepted"] >= 3
        assert _state_err(x, so32.sample_ode_from_base(base, None, "dopri5", dict(opts))) < ADAPT_TOL, list(opts)
    lp = sm.log_prob(base.to(DEV) * 0.5, options={"min_step": 1e-6, "norm": lambda state: max(c.abs().max() for c in state)})
    assert "chunks" not in sm.last_solver_stats
    ref = so32.log_prob(base * 0.5, None, "dopri5", {"min_step": 1e-6, "norm": lambda state: max(c.abs().max() for c in state)}, "exact", None)
    assert _logp_err(lp, ref) < ADAPT_TOL


def test_diverging_vp_reverse_flow_is_pinned_on_both_sides(monkeypatch):
    """Round 2's red run (`r02a`): PopulationModelDiffusion.forward with the reference's default method on a RANDOM-INIT
    VP score network -- the reverse flow grows without bound, the error estimate turns non-finite and torchdiffeq's
    next attempt asserts `underflow in dt nan`.  Product (either controller) and oracle raise exactly that, after the
    same number of attempted and accepted steps.  (What the real torchdiffeq does there is unpinned like the rest of the
    stepper; this pins the product to the oracle instead of dropping the case.)"""
    from oracle import flowfusion_oracle as O
    from tests.test_gpu_full_configs import _population
    for conditional in (False, True):
        pm, po32, _ = _population(conditional, "VPSDE", True, False, "dopri5", None, 161)
        torch.manual_seed(5)
        base = torch.randn(120, 5)
        cond = torch.randn(120, 3) * 1.5 + 0.3 if conditional else None
        args = () if cond is None else (cond.to(DEV),)

        def product():
            with pytest.raises(RuntimeError, match="underflow in dt nan") as ei:
                pm(base.to(DEV), *args)
            return None, ei.value.solver_stats
        (_, sd), (_, sh) = _both(monkeypatch, product)
        with pytest.raises(AssertionError, match="underflow in dt nan"):
            po32.forward(base, cond)
        so = dict(O.last_adaptive_stats)
        assert _same_counts(sd, sh) and _same_counts(sd, so), (sd, sh, so)
        assert sd["attempts"] >= 2


def test_split_precision_under_the_device_controller(monkeypatch):
    """precision="bf16x2" / "bf16x3": the split kernels honour the gate word and run the same device loop."""
    for prec in ("bf16x2", "bf16x3"):
        sm, so32, _ = _seeded_score_model(9, 3, [128, 100, 128], "VESDE", False, 71)
        sm.precision = prec
        torch.manual_seed(6)
        base, cond = torch.randn(300, 9), torch.randn(300, 3)

        def run():
            x, _ = sm.sample_ode_from_base(base.to(DEV), conditional=cond.to(DEV))
            return x, dict(sm.last_solver_stats)
        (a, sa), (b, sb) = _both(monkeypatch, run)
        assert abs(sa["attempts"] - sb["attempts"]) <= 1 and _state_err(a, b.cpu()) < 2e-5, (prec, sa, sb)
        assert _state_err(a, so32.sample_ode_from_base(base, cond, "dopri5", None)) < ADAPT_TOL


def test_large_batch_default_log_prob_runs_in_one_or_two_chunks():
    """2^16 points of BASELINE config 2's network, default arguments: the device loop needs at most one extra look at the
    state, and repeats bit for bit."""
    sm, _, _ = _seeded_score_model(16, 0, [256] * 4, "VPSDE", True, 5)
    sm.hutch = True
    g = torch.Generator(device=DEV).manual_seed(1)
    x0 = torch.randn(1 << 16, 16, device=DEV, generator=g) * 0.8
    torch.manual_seed(9)
    a = sm.log_prob(x0)
    st = dict(sm.last_solver_stats)
    torch.manual_seed(9)
    b = sm.log_prob(x0)
    assert torch.equal(a, b) and st == sm.last_solver_stats and st["chunks"] <= 3 and torch.isfinite(a).all()


@pytest.mark.parametrize("case", ["VESDE", "VPSDE", "SUBVPSDE", "flow"])
def test_rows_written_by_the_controller_equal_the_host_rows(case):
    """The evaluation table the controller kernel writes for the next attempt, read back after every attempted step, against
    the rows the host controller builds for the same (t, dt) (fused.make_step + the torch schedule): integer words, stage
    coefficients and tail rows bit for bit, a_e / b_e to two ulps, c1 to the rounding of an 8-term fp32 sum (device vs
    torch's vectorised sin / cos / exp / pow)."""
    from flowfusion_amd import adaptive, device_adaptive
    from flowfusion_amd import flow as Fm
    torch.manual_seed(12)
    if case == "flow":
        model = Fm.ConditionalODEFlow(5, 2, [96, 128]).eval().to(DEV)
        x, cond = torch.randn(300, 5, device=DEV), torch.randn(300, 2, device=DEV)
        run = lambda: model.sample(x, cond, atol=1e-7, rtol=1e-7)
        net = model._net()
        first = net.first_layer_cpu()
        sched, sign = (lambda tr: model._schedule(tr, first)), -1.0
    else:
        sm, _, _ = _seeded_score_model(5, 2, [96, 128], case, case == "VPSDE", 55)
        x, cond = torch.randn(300, 5, device=DEV), torch.randn(300, 2, device=DEV)
        net = sm._net()
        host = sm._schedule_inputs()
        if case == "VESDE":
            run = lambda: sm.sample_ode_from_base(x, conditional=cond, atol=1e-7, rtol=1e-7)
            sched, sign = (lambda tr: sm._schedule(tr, "ode", host)[:3]), -1.0
        else:       # (a random-init VP reverse flow diverges: the forward solve of log_prob walks the same schedule upwards)
            sm.hutch = True
            run = lambda: sm.log_prob(x * 0.5, conditional=cond, atol=1e-6, rtol=1e-6)
            sched, sign = (lambda tr: sm._schedule(tr, "ode", host)[:3]), 1.0
    device_adaptive.TRACE, device_adaptive.TRACE_ROWS = [], True
    try:
        run()
        trace = device_adaptive.TRACE
    finally:
        device_adaptive.TRACE, device_adaptive.TRACE_ROWS = None, False
    assert len(trace) >= 3
    # the host controller's rows for the same step: capture what make_step hands to the launcher
    captured = {}

    def launcher(y, k1, kl1, lp0, rows, n_aux, first_, count, **kw):
        captured["rows"] = rows.clone()
        return torch.zeros(n_aux, 1, 5), None
    step = net.make_step(sched, sign, 0, "cpu", cond=None, probe=None, launcher=launcher)      # (rows do not depend on the mode)
    helper = adaptive.Dopri5(step, False, 1e-5, 1e-5, None)
    width = net.plan(0).width
    checked = 0
    for (n_att, n_acc, t, dt, ratio, rows_dev) in trace[:-1]:          # the last entry is the finished solve: no next attempt
        helper._attempt(t, dt, t + dt, torch.zeros(1, 5), None, torch.zeros(1, 5), None)
        rows_host = captured["rows"]                                    # [6 + 2, 32 + width]
        dev = rows_dev[: rows_host.shape[0]]
        hi, di = rows_host.view(torch.int32), dev.view(torch.int32)
        assert torch.equal(hi[:, 3:6], di[:, 3:6])                      # flags / use_y bits, slots, noise index
        assert torch.equal(rows_host[:, 8:24], dev[:, 8:24])            # stage and tail coefficients: bit for bit
        assert torch.equal(rows_host[:, 24:32], dev[:, 24:32]) and torch.equal(rows_host[6:, 32:], dev[6:, 32:])
        # (sub-VP: sigma = 1 - exp(-x) and 1 - decay cancel at small t, so one ulp of `exp` shows up amplified in b_e)
        torch.testing.assert_close(dev[:6, :2], rows_host[:6, :2], rtol=2e-5 if case == "SUBVPSDE" else 5e-7, atol=0)
        torch.testing.assert_close(dev[:6, 32:32 + width], rows_host[:6, 32:32 + width], rtol=2e-6, atol=3e-6)
        checked += 1
    assert checked >= 2


@pytest.mark.parametrize("B", [1, 3, 17])
def test_tiny_batches_on_the_device_controller(B, monkeypatch):
    """One, three, seventeen rows (unaligned array ends, a single tile, one reduction block): sampling and exact-trace
    log-density under both controllers and against the oracle; an empty batch returns empty tensors."""
    sm, so32, _ = _seeded_score_model(3, 2, [64, 64], "VESDE", False, 404)
    torch.manual_seed(B)
    base, cond = torch.randn(B, 3), torch.randn(B, 2)

    def sample():
        x, _ = sm.sample_ode_from_base(base.to(DEV), conditional=cond.to(DEV))
        return x, dict(sm.last_solver_stats)
    (a, sa), (b, sb) = _both(monkeypatch, sample)
    # (with a handful of rows nothing averages the step control: the two controllers' step sizes differ in the seventh digit
    # and the solutions by a fraction of rtol = 1e-4 -- the bar between them is the solver tolerance's, not 1e-5)
    assert a.shape == (B, 3) and _same_counts(sa, sb) and _state_err(a, b.cpu()) < ADAPT_TOL
    assert _state_err(a, so32.sample_ode_from_base(base, cond, "dopri5", None)) < ADAPT_TOL

    def logp():
        return sm.log_prob(base.to(DEV) * 0.5, conditional=cond.to(DEV)), dict(sm.last_solver_stats)
    (a, sa), (b, sb) = _both(monkeypatch, logp)
    assert a.shape == (B, 1) and _same_counts(sa, sb) and _logp_err(a, b.cpu()) < ADAPT_TOL
    assert _logp_err(a, so32.log_prob(base * 0.5, cond, "dopri5", {"min_step": 1e-6}, "exact", None)) < ADAPT_TOL
    if B == 1:
        e, _ = sm.sample_ode_from_base(base[:0].to(DEV), conditional=cond[:0].to(DEV))
        assert e.shape == (0, 3) and sm.last_solver_stats["attempts"] == 0


# ---- kernel-level guards ----------------------------------------------------------------------------------------------
def _raw_launch(sm, x, table, gate=None, stage_slots=0, mode=0, status=None):
    from flowfusion_amd import _native
    net = sm._net()
    plan = net.plan(mode)
    wpack = net.wpack(x.device, mode)
    out = torch.full_like(x, -123.0)
    a = _native.OdeArgs()
    a.x_in, a.x_out, a.wpack, a.etab = x.data_ptr(), out.data_ptr(), wpack.data_ptr(), table.data_ptr()
    a.batch, a.n_evals, a.mode, a.stage_slots = x.shape[0], table.shape[0], mode, stage_slots
    a.status = 0 if status is None else status.data_ptr()
    a.gate = 0 if gate is None else gate.data_ptr()
    rc = _native.lib().ff_mlp_ode_launch(ctypes.byref(plan), ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return rc, out


@pytest.mark.parametrize("prec", ["f32", "bf16x2"])
def test_gate_word_makes_a_launch_a_no_op(prec):
    sm, _, _ = _seeded_score_model(6, 0, [128, 128], "VPSDE", True, 3)
    sm.precision = prec
    x = torch.randn(500, 6, device=DEV)
    table = sm._ode_table(torch.tensor([1.0, 1e-3]), "rk4", {"step_size": 0.1}, 0).to(DEV)
    closed = torch.zeros(1, dtype=torch.int32, device=DEV)
    rc, out = _raw_launch(sm, x, table, gate=closed, stage_slots=4)
    assert rc == 0 and bool((out == -123.0).all())                      # nothing was written
    rc, out = _raw_launch(sm, x, table, gate=torch.ones(1, dtype=torch.int32, device=DEV), stage_slots=4)
    want, _ = sm.sample_ode_from_base(x, method="rk4", options={"step_size": 0.1})
    assert rc == 0 and torch.equal(out, want)


def test_rows_naming_a_slot_off_chip_are_refused():
    """the stage slot of a row was used unchecked.  f32 kernels keep FF_MAX_SLOTS = 7; the four-slot twins
    of the split family keep 4 and are chosen from the caller's `stage_slots` promise: a dopri5 table behind a promise of
    4 used to overwrite the parked stage input and the state in LDS.  Now the row is refused and the status word says so;
    plans with four slots reject a larger promise before any launch."""
    from flowfusion_amd import _native
    sm, _, _ = _seeded_score_model(6, 0, [128, 128], "VPSDE", True, 3)
    x = torch.randn(300, 6, device=DEV)
    good = sm._ode_table(torch.tensor([1.0, 1e-3]), "euler", {"step_size": 0.25}, 0)
    bad = good.clone()
    bad.view(torch.int32)[1, 4] = 7                                      # slot 7 of 0..6
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    rc, _ = _raw_launch(sm, x, good.to(DEV), status=status)
    assert rc == 0 and int(status.item()) == 0
    rc, _ = _raw_launch(sm, x, bad.to(DEV), status=status)
    assert rc == 0 and int(status.item()) & _native.STATUS_BAD_SLOT
    # the four-slot twin behind a wrong promise
    sm.precision = "bf16x2"
    dp = sm._ode_table(torch.tensor([1.0, 1e-3]), "dopri5_fixed", {"step_size": 0.25}, 0).to(DEV)      # slots 0..5
    status.zero_()
    rc, lied = _raw_launch(sm, x, dp, stage_slots=4, status=status)
    flagged = bool(int(status.item()) & _native.STATUS_BAD_SLOT)
    status.zero_()
    rc2, out = _raw_launch(sm, x, dp, stage_slots=6, status=status)      # an honest promise: the seven-slot kernel
    want, _ = sm.sample_ode_from_base(x, method="dopri5_fixed", options={"step_size": 0.25})
    assert rc == 0 and rc2 == 0 and int(status.item()) == 0 and torch.equal(out, want)
    # a library with the four-slot twins (FF_BUILD_FULL) serves the false promise with a twin, which refuses rows 4 and 5; the
    # default library has no twin, the seven-slot kernel serves the launch and the promise did not matter -- never a silent
    # wrong answer in between
    assert flagged or torch.equal(lied, want)
    # plans that keep four slots (17-32 dimensions) reject a larger promise
    sw, _, _ = _seeded_score_model(20, 0, [128, 128], "VPSDE", True, 4)
    sw.precision = "bf16x2"
    xw = torch.randn(64, 20, device=DEV)
    tw = sw._ode_table(torch.tensor([1.0, 1e-3]), "rk4", {"step_size": 0.25}, 0).to(DEV)
    rc, _ = _raw_launch(sw, xw, tw, stage_slots=7)
    assert rc == _native.FF_ERR_UNSUPPORTED
    rc, _ = _raw_launch(sw, xw, tw, stage_slots=4)
    assert rc == 0


def test_global_step_control_over_ranks(tmp_path):
    """distributed.global_step_control: torchdiffeq's step size comes from a norm over the WHOLE batch, so an adaptive
    solve sharded over ranks has one real exchange step -- the sums of squares behind every norm (ff_adapt_buffers.exchange:
    reduce kernel, all-reduce of 8 doubles enqueued by the host hook, controller kernel).  Three ranks on cuda:0 over gloo
    (RCCL refuses several ranks on one device; a fresh child process tree): under the context every rank attempts and
    accepts exactly the steps of the whole-batch solve and its rows agree with that solve to rounding, for the score
    model's sampler, its exact-trace log_prob, a flow and the host step controller; without it the ranks' own norms
    give different step sequences (the shards are built to differ)."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    W = 3
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(W), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(root / "tests" / "_global_control_worker.py")]
    import os
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(root),
                       env=dict(os.environ, FF_RESULT_DIR=str(tmp_path)))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    recs = [json.loads(p.read_text()) for p in sorted(tmp_path.glob("rank*.json"))]
    assert sorted(x["rank"] for x in recs) == list(range(W))
    differs = 0
    for rec in recs:
        assert rec["empty_raises"] is True
        for name in ("sample_ve_2d", "log_prob_ve_2d_exact", "flow_sample_8d", "sample_ve_2d_host_controller"):
            c = rec[name]
            for k in ("attempts", "accepted"):
                assert c["global"][k] == c["whole"][k], (name, rec["rank"], c)
            assert c["err_global"] < 2e-5, (name, rec["rank"], c)
            assert c["err_alone"] < 2e-3, (name, rec["rank"], c)     # a shard on its own: other steps, same answer to the solver tolerances
            differs += c["alone"]["attempts"] != c["whole"]["attempts"] or c["err_alone"] > 10 * max(c["err_global"], 1e-7)
        assert rec["sample_ode_sharded"]["span_ok"] and rec["sample_ode_sharded"]["err"] < 2e-5, rec
        lh = rec["log_prob_sharded_hutch"]
        assert lh["span_ok"] and lh["same_steps"] and lh["err"] < 2e-5, lh
        for name in ("flow_sample_sharded", "flow_log_prob_sharded_hutch"):
            c = rec[name]
            assert c["span_ok"] and c["same_steps"] and c["err"] < 2e-5, (name, c)
        assert rec["flow_sample_sharded_gathered"] is True and rec["too_few_rows_raises"] is True
    assert differs > 0        # the exchange is what made the step sequences equal


def test_cooperative_twin_is_deterministic_when_the_card_is_shared(tmp_path):
    """Regression (round 3), two races of the cooperative twin that only show when its wavefronts do not run in step:
    (1) the four wavefronts of a workgroup share the Runge-Kutta stage slots, and a wavefront that started late
    zero-filled them AFTER another had stored the caller's first stage (k1_in, every adaptive attempt) and before that one
    read it back; (2) the activation exchange alternated its two LDS buffers per evaluation, so with an odd number of
    hidden layers one evaluation ended and the next began on the same buffer, one barrier apart.  Alone on the card the
    wavefronts start and run together and nothing shows; with three processes on the card nearly every repeat of an
    adaptive solve gave another step sequence.  Three processes repeat a default-argument log_prob at a cooperative-twin
    batch (3001 x 16-d, both controllers) and the notebook model's sampler (3 x 128: odd): one fingerprint each, the same
    in every process."""
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
    W = 3
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(W), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(root / "tests" / "_contention_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(root),
                       env=dict(os.environ, FF_RESULT_DIR=str(tmp_path)))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    recs = [json.loads(p.read_text()) for p in sorted(tmp_path.glob("contention*.json"))]
    assert len(recs) == W
    for rec in recs:
        assert len(rec) == 3 and sorted(rec.values()) == [25, 25, 25], rec      # one fingerprint per case
        assert rec.keys() == recs[0].keys()


def test_two_host_threads_on_two_streams():
    """include/flowfusion_amd.h: "safe to call from several host threads on different streams".  Two threads, each on a
    stream of its own, repeat different default-argument solves at the same time (the norm workspace is per stream, the work
    buffers per call); every result equals the one the solve gives alone on the default stream, bit for bit."""
    import threading
    from flowfusion_amd.diffusion import MLP, VESDE, VPSDE, ScoreModel
    torch.manual_seed(0)
    a = ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).eval().to(DEV)
    torch.manual_seed(1)
    b = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True, hutchinson=True).eval().to(DEV)
    za = torch.randn(5000, 2, device=DEV) * 3
    xb = torch.randn(3001, 16, device=DEV) * 0.8
    ref_a, _ = a.sample_ode_from_base(za)
    ref_b = b.log_prob(xb, probe="philox", seed=4)
    torch.cuda.synchronize()
    bad, errors = [], []

    def work(fn, ref, tag):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for i in range(15):
                    r = fn()
                    s.synchronize()
                    if not torch.equal(r, ref):
                        bad.append((tag, i))
        except Exception as exc:          # noqa: BLE001
            errors.append((tag, repr(exc)))

    ts = [threading.Thread(target=work, args=(lambda: a.sample_ode_from_base(za)[0], ref_a, "sample")),
          threading.Thread(target=work, args=(lambda: b.log_prob(xb, probe="philox", seed=4), ref_b, "log_prob"))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not errors and not bad, (errors, bad)


def test_default_adaptive_solves_against_scipys_integrator():
    """An anchor that is nobody's restatement of torchdiffeq: scipy's own RK45 (`solve_ivp`, rtol 1e-10) integrates the
    ORACLE's float64 right-hand side -- the reference's formulas, pinned by its fixtures -- to convergence; the product's
    adaptive solves of the same problem (dopri5 on the device controller, the other pairs, dopri8 stage by stage) must land
    within their tolerance of that answer, for sampling (decreasing span) and for the exact-trace log-density."""
    import numpy as np
    from scipy.integrate import solve_ivp
    sm, _, so64 = _seeded_score_model(4, 0, [64, 64], "VESDE", False, 333)
    torch.manual_seed(1)
    B, D = 12, 4
    base = torch.randn(B, D)
    eps = float(torch.tensor(float(sm.sde.epsilon), dtype=torch.float32))

    def f_state(t, y):
        x = torch.from_numpy(y.reshape(B, D).copy())
        (xd,) = so64.rhs(torch.tensor(t, dtype=torch.float64), (x,), None, None)
        return xd.reshape(-1).numpy()
    z = (base.double() * float(sm.sde.sigma_max)).reshape(-1).numpy()
    conv = solve_ivp(f_state, (1.0, eps), z, method="RK45", rtol=1e-10, atol=1e-12)
    assert conv.success
    want = torch.from_numpy(conv.y[:, -1].reshape(B, D)).float()
    for method, tol, bar in (("dopri5", 1e-6, 2e-5), ("bosh3", 1e-5, 5e-4), ("dopri8", 1e-6, 2e-5), ("dopri5", 1e-4, 2e-3)):
        x, _ = sm.sample_ode_from_base(base.to(DEV), method=method, atol=tol, rtol=tol)
        assert _state_err(x, want) < bar, (method, tol, _state_err(x, want))

    x0 = torch.randn(B, D) * 0.5

    def f_logp(t, y):
        x = torch.from_numpy(y[: B * D].reshape(B, D).copy())
        xd, div = so64.rhs(torch.tensor(t, dtype=torch.float64), (x, torch.zeros(B, 1, dtype=torch.float64)), None, "exact", None)
        return np.concatenate([xd.detach().reshape(-1).numpy(), div.detach().reshape(-1).numpy()])
    y0 = np.concatenate([x0.double().reshape(-1).numpy(), np.zeros(B)])
    conv = solve_ivp(f_logp, (eps, 1.0), y0, method="RK45", rtol=1e-10, atol=1e-12)
    assert conv.success
    xT = torch.from_numpy(conv.y[: B * D, -1].reshape(B, D))
    from oracle import flowfusion_oracle as O
    want_lp = torch.from_numpy(conv.y[B * D:, -1]).reshape(B, 1) + O.normal_log_prob(xT, so64.sde.prior_scale()).sum(1, keepdim=True)
    for method, tol, bar in (("dopri5", 1e-6, 2e-5), ("dopri5", 1e-4, 2e-3), ("adaptive_heun", 1e-5, 2e-3)):
        lp = sm.log_prob(x0.to(DEV), method=method, atol=tol, rtol=tol)
        assert _logp_err(lp, want_lp.float()) < bar, (method, tol, _logp_err(lp, want_lp.float()))
