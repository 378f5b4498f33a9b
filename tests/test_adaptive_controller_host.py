"""The device-side adaptive controller's arithmetic (csrc/ff_adapt_logic.h), run on the host through
ff_adapt_host_row / ff_adapt_host_transition, against the two statements it must agree with:

* the torch schedule code of the front ends (``ScoreModel._schedule_on_host``, ``_FlowBase._schedule`` -- themselves
  pinned to the reference's SDE classes by tests/golden/sde_schedules.npz, test_oracle_golden.py), row by row;
* the host controller ``adaptive.Dopri5`` (the restatement of torchdiffeq's step control the GPU tests compare with the
  oracle), decision by decision, on scripted error ratios -- including the clamps, the NaN path and the three assertions.

No GPU: the same header is compiled into the kernels (ff_adaptive.hip); tests/test_gpu_device_adaptive.py runs those.
"""
import ctypes
import math

import pytest
import torch

from flowfusion_amd import _native, adaptive, device_adaptive
from flowfusion_amd import diffusion as D
from flowfusion_amd import flow as F


def _config(spec, sign=1.0, method="dopri5", rtol=1e-5, atol=1e-5, options=None):
    keep = []
    return device_adaptive.build_config(spec, sign, method, rtol, atol, options, keep), keep


def _score_spec(sm):
    """ScheduleSpec with HOST tensors (ff_adapt_host_row reads host pointers)."""
    spec = sm._device_schedule("cpu")
    assert spec is not None
    return spec


@pytest.mark.parametrize("sde_name", ["VESDE", "VPSDE", "SUBVPSDE"])
@pytest.mark.parametrize("no_sigma", [False, True])
def test_schedule_rows_match_the_torch_schedule(sde_name, no_sigma, built_library):
    torch.manual_seed(3)
    sm = D.ScoreModel(D.MLP(5, 2, 8, [48, 64]), getattr(D, sde_name)(), no_sigma=no_sigma).eval()
    spec = _score_spec(sm)
    H = 48
    eps = float(sm.sde.epsilon)
    ts = torch.tensor([eps, 0.003, 0.05, 0.2, 0.5, 0.77, 0.999, 1.0], dtype=torch.float32)
    a, b, c1, _ = sm._schedule(ts, "ode")
    for sign in (1.0, -1.0):
        cfg, keep = _config(spec, sign)
        for i, t in enumerate(ts.tolist()):
            ao, bo = ctypes.c_float(), ctypes.c_float()
            c1o = torch.empty(H)
            rc = built_library.ff_adapt_host_row(ctypes.byref(cfg), t, ctypes.byref(ao), ctypes.byref(bo), c1o.data_ptr())
            assert rc == 0
            # libm vs torch's vectorised transcendentals: an ulp or two
            assert ao.value == pytest.approx(sign * float(a[i]), rel=2e-6, abs=1e-30), (sde_name, t)
            assert bo.value == pytest.approx(sign * float(b[i]), rel=4e-6), (sde_name, t)
            # c1 sums 8 products of sin / cos of arguments up to ~300 rad: absolute agreement at the fp32 rounding of the sum
            torch.testing.assert_close(c1o, c1[i], rtol=1e-5, atol=2e-5)


def test_flow_rows_match_the_torch_schedule(built_library):
    torch.manual_seed(4)
    f = F.ConditionalODEFlow(6, 3, [40, 40]).eval()
    net = f._net()
    w0t, b0 = net.time_columns("cpu", 6, 7)
    spec = device_adaptive.ScheduleSpec(_native.SCHED_FLOW, (0.0, 0.0, 0.0), True, None, 0.0, w0t, b0)
    ts = torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0])
    a, b, c1 = f._schedule(ts)
    cfg, keep = _config(spec, -1.0)
    for i, t in enumerate(ts.tolist()):
        ao, bo = ctypes.c_float(), ctypes.c_float()
        c1o = torch.empty(40)
        assert built_library.ff_adapt_host_row(ctypes.byref(cfg), t, ctypes.byref(ao), ctypes.byref(bo), c1o.data_ptr()) == 0
        assert ao.value == 0.0 and bo.value == -1.0
        assert torch.equal(c1o, c1[i])                      # one product and one sum per element: bit for bit


class _Scripted(adaptive.Dopri5):
    """adaptive.Dopri5 with the kernel launches and reductions replaced by a script of (ratio, non-finite) pairs; records
    the (t, dt) of every attempt."""

    def __init__(self, script, method, options, first_step_norms=None):
        super().__init__(step=None, has_lp=False, rtol=1e-5, atol=1e-5, options=options, method=method)
        self.script = list(script)
        self.attempts = []
        self.init_norms = first_step_norms

    def _deriv(self, t, y, lp, k1=None, kl1=None, h=None):
        return torch.zeros(1), None

    def _attempt(self, t0, dt, t1, y, lp, f0, fl0):
        self.attempts.append((t0, dt))
        return [torch.zeros(1)] * 4, None

    def _norms(self, terms, check=None):
        if not self.script:
            raise IndexError("script exhausted")
        return self.script.pop(0)

    def _interp(self, interp, t):
        return None, None


def _run_host(script, method, options, t0, t_end, first_step):
    opts = dict(options or {})
    opts["first_step"] = first_step
    s = _Scripted(script, method, opts)
    err = None
    try:
        s.integrate(t0, t_end, torch.zeros(1), None)
    except RuntimeError as e:
        err = str(e)
    except IndexError:
        err = "script exhausted"
    return s, err


def _run_device_logic(lib, script, method, options, t0, t_end, first_step):
    """The same walk through ff_adapt_host_transition."""
    w0t, b0 = torch.zeros(4, 1), torch.zeros(4)
    spec = device_adaptive.ScheduleSpec(_native.SCHED_FLOW, (0.0, 0.0, 0.0), True, None, 0.0, w0t, b0)
    opts = dict(options or {})
    opts["first_step"] = first_step
    cfg, keep = _config(spec, 1.0, method, options=opts)
    st = _native.AdaptState()
    st.t, st.t_prev, st.t_end, st.active = t0, t0, t_end, 1
    attempts = []
    go = lib.ff_adapt_host_transition(ctypes.byref(cfg), ctypes.byref(st), 4, None)          # first_step given
    script = list(script)
    while go == 1:
        attempts.append((st.t, st.dt))
        if not script:
            return st, attempts, "script exhausted"
        ratio, bad = script.pop(0)
        norms = (ctypes.c_float * 2)(ratio, 1.0 if bad else 0.0)
        go = lib.ff_adapt_host_transition(ctypes.byref(cfg), ctypes.byref(st), 3, norms)
    msg = {0: None, 1: f"underflow in dt {st.dt}", 2: "non-finite values in state `y`",
           3: f"max_num_steps exceeded ({st.n_steps}>={cfg.max_num_steps})"}[st.error]
    return st, attempts, msg


SCRIPTS = {
    "accepts_and_rejects": ([(0.5, False), (2.0, False), (0.9, False), (1.0, False), (1.7, False), (0.01, False), (0.3, False)] * 6, {}),
    "zero_ratio_grows_tenfold": ([(0.0, False)] * 8, {}),
    "min_step_forces_acceptance": ([(50.0, False)] * 400, {"min_step": 0.02}),
    "max_step_clamps_growth": ([(1e-6, False)] * 60, {"max_step": 0.03}),
    "nan_ratio_underflows_next": ([(0.4, False), (float("nan"), False)], {}),
    "nonfinite_state_on_accept": ([(0.4, False), (0.5, True)], {}),
    "nonfinite_state_rejected_is_fine": ([(0.4, False), (3.0, True), (0.2, False)] + [(0.3, False)] * 40, {}),
    "max_num_steps": ([(0.9, False)] * 50, {"max_num_steps": 5}),
    "shrinks_to_underflow": ([(1e30, False)] * 2000, {}),
}


@pytest.mark.parametrize("method", ["dopri5", "bosh3", "adaptive_heun"])
@pytest.mark.parametrize("name", list(SCRIPTS))
def test_transitions_match_the_host_controller(name, method, built_library):
    script, options = SCRIPTS[name]
    # the norms reach either controller as fp32 values (the reduction kernel writes floats)
    script = [(float(torch.tensor(r, dtype=torch.float32)), bad) for r, bad in script]
    t0, t_end, first = 1e-3, 1.0, 0.013
    host, herr = _run_host(script, method, options, t0, t_end, first)
    st, attempts, derr = _run_device_logic(built_library, script, method, options, t0, t_end, first)
    assert derr == herr, (name, herr, derr)
    assert len(attempts) == len(host.attempts)
    for (ta, da), (tb, db) in zip(host.attempts, attempts):
        assert ta == tb and da == db, (name, (ta, da), (tb, db))                 # float64, bit for bit
    assert st.n_accepted == host.n_accepted
    if herr is None:
        assert st.done == 1 and st.active == 0 and st.n_attempts == host.n_attempts and st.t >= t_end
    if name == "shrinks_to_underflow":
        assert herr.startswith("underflow in dt") and host.n_accepted == 0
    if name == "max_num_steps":
        assert herr == "max_num_steps exceeded (5>=5)"
    if name == "nan_ratio_underflows_next":
        assert herr == "underflow in dt nan"


@pytest.mark.parametrize("d0,d1,d2n", [(3.0, 40.0, 7.0), (1e-7, 5.0, 1.0), (2.0, 1e-9, 0.5), (0.5, 1e-16, 0.0), (800.0, 1.5e4, 3e5)])
@pytest.mark.parametrize("order", ["dopri5", "bosh3"])
def test_initial_step_rule_matches_the_host_controller(d0, d1, d2n, order, built_library):
    """torchdiffeq `_select_initial_step` as adaptive.Dopri5 computes it from the three norms."""
    tab = adaptive.TABLEAUX[order]
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    h0 = float(torch.as_tensor(abs(h0), dtype=torch.float32))
    d2 = abs(float(torch.as_tensor(d2n, dtype=torch.float32)) / h0)
    d1f = float(torch.as_tensor(d1, dtype=torch.float32))
    if d1f <= 1e-15 and d2 <= 1e-15:
        h1 = max(1e-6, h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1f, d2)) ** (1.0 / float(tab.order))
    want = min(100 * h0, abs(h1))
    w0t, b0 = torch.zeros(4, 1), torch.zeros(4)
    spec = device_adaptive.ScheduleSpec(_native.SCHED_FLOW, (0.0, 0.0, 0.0), True, None, 0.0, w0t, b0)
    cfg, keep = _config(spec, 1.0, order)
    st = _native.AdaptState()
    st.t, st.t_end, st.active = 0.0, 1.0, 1
    n1 = (ctypes.c_float * 2)(d0, d1)
    assert built_library.ff_adapt_host_transition(ctypes.byref(cfg), ctypes.byref(st), 1, n1) == 1
    assert st.h0 == h0
    n2 = (ctypes.c_float * 2)(d2n, 0.0)
    assert built_library.ff_adapt_host_transition(ctypes.byref(cfg), ctypes.byref(st), 2, n2) == 1
    assert st.dt == want and math.isfinite(st.dt)


def test_config_rejects_what_the_host_controller_rejects(built_library):
    spec = device_adaptive.ScheduleSpec(_native.SCHED_FLOW, (0.0, 0.0, 0.0), True, None, 0.0, torch.zeros(4, 1), torch.zeros(4))
    with pytest.raises(NotImplementedError, match="dopri8"):
        _config(spec, 1.0, "dopri8")
    with pytest.raises(NotImplementedError, match="step_t"):
        _config(spec, 1.0, "dopri5", options={"step_t": torch.tensor([0.5])})
    cfg, _ = _config(spec, -1.0, "fehlberg2", rtol=1e-3, atol=1e-4, options={"min_step": 1e-6})
    assert (cfg.n_stages, cfg.order, cfg.sign, cfg.min_step) == (3, 2, -1.0, 1e-6) and math.isnan(cfg.first_step)
    assert cfg.max_num_steps == 2 ** 31 - 1 and cfg.max_step == float("inf")


# ---- a whole solve, walked on the CPU with the device path's arithmetic ---------------------------------------------------
def _walk(lib, sm, net, mode, sign, x, lp, t0, t_end, rtol, atol, options, cond=None, probe=None):
    """An adaptive solve as ff_mlp_ode_adaptive runs it, on the CPU: evaluation rows from ff_adapt_host_row (the schedule
    arithmetic the controller kernel compiles), every decision from ff_adapt_host_transition (the control law it
    compiles), the fused launches from the kernel-semantics emulator, norms / commit / dense output as the kernels compute
    them (adaptive.Dopri5's CPU statements of the same).  Returns (y, lp, state)."""
    from tests import _emulator as E
    from flowfusion_amd.fused import MODE_STATE
    spec = sm._device_schedule("cpu")
    keep = []
    cfg = device_adaptive.build_config(spec, sign, "dopri5", rtol, atol, options, keep)
    H = spec.w0t.shape[0]

    def sched(t_real):                       # (a, b, c1) of fused.make_step's contract, from the C rows
        a, b, c1 = torch.empty_like(t_real), torch.empty_like(t_real), torch.empty(t_real.numel(), H)
        for i, t in enumerate(t_real.tolist()):
            ao, bo = ctypes.c_float(), ctypes.c_float()
            assert lib.ff_adapt_host_row(ctypes.byref(cfg), t, ctypes.byref(ao), ctypes.byref(bo), c1[i].data_ptr()) == 0
            a[i], b[i] = sign * ao.value, sign * bo.value          # the C row carries the reversal sign already
        return a, b, c1
    plan = _native.plan_words(net.plan(mode))
    wpack = net.wpack("cpu", mode)
    launcher = lambda y, k1, kl1, lp0, etab, n_aux, first, count: E.emulate_step(
        plan, wpack, etab, y, cond, probe, k1, kl1, lp0, mode, n_aux, first, count)
    step = net.make_step(sched, sign, mode, "cpu", cond=cond, probe=probe, launcher=launcher)
    has_lp = mode != MODE_STATE
    h = adaptive.Dopri5(step, has_lp, rtol, atol, options)              # its kernels' CPU statements; NOT its control flow
    f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))
    parts = lambda *ps: [p for p in ps if p is not None]
    st = _native.AdaptState()
    st.t, st.t_prev, st.t_end, st.active = t0, t0, t_end, 1
    y, f0, fl0 = x, None, None
    f0, fl0 = h._deriv(t0, y, lp)
    ys, fs = parts(y, lp), parts(f0, fl0)
    scale = [atol + p.abs() * rtol for p in ys]
    d0 = f32(adaptive._mixed_norm([a / s for a, s in zip(ys, scale)]))
    d1 = f32(adaptive._mixed_norm([a / s for a, s in zip(fs, scale)]))
    assert lib.ff_adapt_host_transition(ctypes.byref(cfg), ctypes.byref(st), 1, (ctypes.c_float * 2)(d0, d1)) == 1
    f1, fl1 = h._deriv(f32(t0) + st.h0, y, lp, k1=f0, kl1=fl0, h=st.h0)
    d2n = f32(adaptive._mixed_norm([(a - b) / s for a, b, s in zip(parts(f1, fl1), fs, scale)]))
    go = lib.ff_adapt_host_transition(ctypes.byref(cfg), ctypes.byref(st), 2, (ctypes.c_float * 2)(d2n, 0.0))
    last = None
    while go == 1:
        aux, aux_lp = h._attempt(st.t, st.dt, st.t + st.dt, y, lp, f0, fl0)
        y1, f1, ymid, yerr = aux[0], aux[1], aux[2], aux[3]
        terms = [(yerr, None, y, y1)]
        if has_lp:
            terms.append((aux_lp[3], None, lp, aux_lp[0]))
        ratio, bad = h._norms(terms, check=y1)
        before = st.n_accepted
        go = lib.ff_adapt_host_transition(ctypes.byref(cfg), ctypes.byref(st), 3, (ctypes.c_float * 2)(f32(ratio), 1.0 if bad else 0.0))
        if st.n_accepted > before:           # accepted: what the buffers hold for the dense output
            last = (y, y1, ymid, f0, f1, (lp, aux_lp[0], aux_lp[2], fl0, aux_lp[1]) if has_lp else None)
        if st.commit:
            y, f0 = y1, f1
            if has_lp:
                lp, fl0 = aux_lp[0], aux_lp[1]
    if not st.done:
        return None, None, st
    xq = (st.t_end - st.t_prev) / (st.t - st.t_prev)                     # adapt_finish_kernel
    ya, yb, ym, fa, fb, lps = last
    out = h._fit_eval(ya, yb, ym, fa, fb, st.dt_prev, xq)
    out_lp = h._fit_eval(*lps[:3], lps[3], lps[4], st.dt_prev, xq) if has_lp else None
    return out, out_lp, st


@pytest.mark.parametrize("case", ["ve_sample", "vp_hutch", "subvp_exact_cond"])
def test_cpu_walk_of_the_device_path_matches_oracle_and_host_controller(case, built_library):
    from tests._util import max_rel, score_oracle
    from tests.test_host_logic import _cpu_launcher
    from flowfusion_amd.fused import MODE_EXACT, MODE_HUTCH, MODE_STATE
    torch.manual_seed(3)
    meta = {"ve_sample": dict(D=6, C=0, E=8, units=[64, 64], sde="VESDE", sde_kw={}, no_sigma=True),
            "vp_hutch": dict(D=3, C=0, E=8, units=[64, 64], sde="VPSDE", sde_kw={}, no_sigma=True),
            "subvp_exact_cond": dict(D=4, C=2, E=8, units=[64], sde="SUBVPSDE", sde_kw={}, no_sigma=False)}[case]
    sm = D.ScoreModel(D.MLP(meta["D"], meta["C"], meta["E"], meta["units"]), getattr(D, meta["sde"])(), no_sigma=meta["no_sigma"]).eval()
    so = score_oracle(meta, {k: v.detach().clone() for k, v in sm.state_dict().items()}, torch.float64)
    net = sm._net()
    B = 7
    x = torch.randn(B, meta["D"])
    cond = torch.randn(B, meta["C"]) if meta["C"] else None
    eps = float(sm.sde.epsilon)
    rtol = atol = 1e-7                        # well below the bar: see test_adaptive_dopri5_driver_matches_oracle
    TOL = 5e-5
    sched = lambda tr: sm._schedule(tr, "ode")[:3]
    if case == "ve_sample":
        x = x * sm.sde.sigma_max
        y, _, st = _walk(built_library, sm, net, MODE_STATE, -1.0, x, None, -1.0, -float(torch.tensor(eps, dtype=torch.float32)), rtol, atol, None)
        ref = so.sample_ode_from_base((x / sm.sde.sigma_max).double(), None, "dopri5", None, atol, rtol)
        assert st.done == 1 and st.n_accepted >= 3 and max_rel(y, ref, floor=ref.abs().max().item()) < TOL
        host = adaptive.Dopri5(net.make_step(sched, -1.0, MODE_STATE, "cpu", launcher=_cpu_launcher(net, MODE_STATE)), False, rtol, atol, None)
        yh, _ = host.integrate(-1.0, -float(torch.tensor(eps, dtype=torch.float32)), x, None)
    else:
        mode = MODE_HUTCH if case == "vp_hutch" else MODE_EXACT
        e = torch.sign(torch.randn(B, meta["D"])) if mode == MODE_HUTCH else None
        t0 = float(torch.tensor(eps, dtype=torch.float32))
        y, lp, st = _walk(built_library, sm, net, mode, 1.0, x, torch.zeros(B), t0, 1.0, rtol, atol, {"min_step": 1e-9}, cond, e)
        xT, dlp = so.solve_odes_forward(x.double(), None if cond is None else cond.double(), "dopri5", {"min_step": 1e-9},
                                        "hutch" if mode == MODE_HUTCH else "exact", None if e is None else e.double(), atol, rtol)
        assert st.done == 1 and st.n_accepted >= 3
        assert max_rel(y, xT, floor=xT.abs().max().item()) < TOL and max_rel(lp[:, None], dlp, floor=1.0) < TOL
        host = adaptive.Dopri5(net.make_step(sched, 1.0, mode, "cpu", cond=cond, probe=e, launcher=_cpu_launcher(net, mode, cond, e)),
                               True, rtol, atol, {"min_step": 1e-9})
        yh, _ = host.integrate(t0, 1.0, x, torch.zeros(B))
    # the host controller on the same emulated kernels: the same walk, give or take the accepts a last-bit difference of a
    # transcendental flips over ~100 steps (the error ratio feeds back into the step size: DESIGN.md section 2.1)
    slack = max(2, round(0.03 * host.n_attempts))
    assert abs(host.n_attempts - st.n_attempts) <= slack and abs(host.n_accepted - st.n_accepted) <= slack
    assert max_rel(y, yh, floor=yh.abs().max().item()) < TOL


def test_adaptive_entry_point_checks_its_arguments_before_any_hip_call(built_library):
    """ff_mlp_ode_adaptive validates the plan / config / buffers on the host: bad arguments come back as FF_ERR_BADARG or
    FF_ERR_UNSUPPORTED with nothing enqueued (no GPU needed to see that), an empty batch is FF_OK."""
    lib = built_library
    plan = _native.make_plan(4, 0, [64, 64], 0)
    spec = device_adaptive.ScheduleSpec(_native.SCHED_FLOW, (0.0, 0.0, 0.0), True, None, 0.0, torch.zeros(64, 1), torch.zeros(64))
    cfg, keep = _config(spec, 1.0, "dopri5")
    buf = torch.zeros(4096)
    b = _native.AdaptBuffers()
    for name in ("y", "f0", "scratch_x", "etab", "out_y", "state", "norm_workspace"):
        setattr(b, name, buf.data_ptr())
    for j in range(4):
        b.aux[j] = buf.data_ptr()
    b.n_passes = 1
    base = _native.OdeArgs()
    base.wpack, base.mode, base.batch = buf.data_ptr(), 0, 0
    call = lambda p=plan, a=base, c=cfg, bb=b, n=4: lib.ff_mlp_ode_adaptive(
        ctypes.byref(p), ctypes.byref(a), ctypes.byref(c), ctypes.byref(bb), 0.0, 1.0, _native.ADAPT_START, n, None)
    assert call() == _native.FF_OK                                         # empty batch: nothing to do
    assert call(n=-1) == _native.FF_ERR_BADARG
    class copy:          # ctypes structures holding pointers do not pickle-copy: clone the bytes
        @staticmethod
        def copy(st):
            return type(st).from_buffer_copy(bytes(st))
    bad = copy.copy(cfg); bad.n_stages = 1
    assert call(c=bad) == _native.FF_ERR_BADARG
    bad = copy.copy(cfg); bad.n_stages = 8
    assert call(c=bad) == _native.FF_ERR_BADARG
    bad = copy.copy(cfg); bad.sched = 7
    assert call(c=bad) == _native.FF_ERR_BADARG
    bad = copy.copy(cfg); bad.h_real = 65                                  # more first-layer rows than the plan's width
    assert call(c=bad) == _native.FF_ERR_UNSUPPORTED
    bad = copy.copy(cfg); bad.n_tcols = 2                                  # a flow has ONE time column
    assert call(c=bad) == _native.FF_ERR_BADARG
    bad = copy.copy(cfg); bad.sched = _native.SCHED_VE                     # a score schedule needs the embedding frequencies
    assert call(c=bad) == _native.FF_ERR_BADARG
    nb = copy.copy(b); nb.out_y = 0
    assert call(bb=nb) == _native.FF_ERR_BADARG
    nb = copy.copy(b); nb.n_passes = 9
    assert call(bb=nb) == _native.FF_ERR_BADARG
    hb = copy.copy(base); hb.mode = 1                                      # a divergence mode without its arrays
    assert call(a=hb) == _native.FF_ERR_BADARG
