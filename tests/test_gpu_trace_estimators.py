"""GPU tier: Hutch++ / XTrace log-densities with the reference's DEFAULT arguments (adaptive dopri5; flowfusion/diffusion.py
:336-481, probes :703-719, the calls demo_diffusion.ipynb times) on the device-resident adaptive path.

* ff_trace_estimate (csrc/ff_trace.hip) against the torch statement of the estimators (trace_estimators.py, itself pinned
  by the reference's fixtures in the CPU tier) on random Jacobians;
* the adaptive solve with the controller on the device -- fused attempt with every row's Jacobian recorded, one estimate
  launch, one combine launch, controller, commit -- against the host route (one launch per right-hand side, host
  controller; FF_HOST_CONTROLLER=1): SAME attempt / accept counts, same log-densities;
* ... against rounds 1-3's host route proper, whose estimator is the torch statement (FF_TORCH_ESTIMATOR=1): the same
  log-densities to solver tolerance.  NOT the same counts in general: an estimate is a difference of O(D^2) products, its
  fp32 rounding (summation order: rocBLAS matmuls there, index order here) reaches the error ratio at the 1e-3 relative
  level where the step's error is far below tolerance, and the ratio steers every later step
  (profiles/r04/estimator_device_vs_host_route.txt prints one such pair attempt by attempt);
* ... and against the CPU oracle's dopri5 with the same probes.
"""
import pytest
import torch

from tests._util import golden_names, load_golden, max_rel, score_model, score_oracle
from tests.test_gpu_parity import ADAPT_TOL, DEV, _logp_err, _seeded_score_model
from tests.test_trace_estimators import well_posed

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(built_library):
    assert torch.cuda.is_available(), "the gpu tier needs a GPU"


@pytest.mark.parametrize("generic", [False, True])
def test_trace_estimate_kernel_against_the_torch_statement(generic, monkeypatch):
    """Both device kernels of ff_trace.hip -- the LDS-tile path (D <= 16 and the tile fits; compile-time dimensions 1, 2, 3,
    4, 8, 16 and the run-time-dimension instantiation) and the general path (pinned with FF_TRACE_GENERIC=1; what larger
    shapes take anyway) -- over ragged tiles (batches that are no multiple of 64, tiles spanning two evaluation rows)."""
    from flowfusion_amd import _native, trace_estimators as TE
    if generic:
        monkeypatch.setenv("FF_TRACE_GENERIC", "1")
    torch.manual_seed(23)
    for D, r, m, B, n in ((2, 1, 1, 1000, 6), (16, 1, 1, 300, 6), (5, 2, 3, 77, 2), (32, 3, 2, 65, 1), (16, 16, 1, 40, 3),
                          (3, 2, 7, 513, 2), (1, 1, 2, 5, 1), (64, 4, 4, 9, 2), (8, 3, 2, 100, 3), (4, 4, 1, 129, 2),
                          (16, 4, 4, 70, 2), (12, 2, 2, 33, 5)):
        A = torch.randn(n, B, D, D)
        S, G = torch.sign(torch.randn(r, B, D)), torch.sign(torch.randn(m, B, D))
        rep = lambda P: P.unsqueeze(1).expand(P.shape[0], n, B, D).reshape(P.shape[0], n * B, D).double()
        ok = well_posed(S).repeat(n)
        flat = A.reshape(n * B, D, D).double()
        scale = max(1.0, float(A.abs().sum(dim=(2, 3)).max()))
        got = _native.trace_estimate(A.to(DEV), "hutchpp", (S.to(DEV), G.to(DEV))).cpu().reshape(-1).double()
        assert float((got - TE.hutchpp(flat, rep(S), rep(G)))[ok].abs().max()) < 2e-6 * scale, (D, r, m)
        # the host statement of the same arithmetic (the CPU tier checks that one against the reference's fixtures)
        host = _native.trace_estimate(A, "hutchpp", (S, G), host=True).reshape(-1).double()
        assert float((got - host)[ok].abs().max()) < 2e-6 * scale, (D, r, m)
        got = _native.trace_estimate(A.to(DEV), "xtrace", (S.to(DEV),)).cpu().reshape(-1).double()
        assert float((got - TE.xtrace(flat, rep(S)))[ok].abs().max()) < 2e-6 * scale * max(1, r), (D, r)


def _with_probes(monkeypatch, probes):
    from flowfusion_amd import trace_estimators as TE
    queue = [p.to(DEV) for p in probes]
    monkeypatch.setattr(TE, "draw_probes", lambda n, like: queue.pop(0))
    return queue


def _both_routes(monkeypatch, fn):
    monkeypatch.delenv("FF_HOST_CONTROLLER", raising=False)
    dev = fn()
    monkeypatch.setenv("FF_HOST_CONTROLLER", "1")
    host = fn()
    monkeypatch.delenv("FF_HOST_CONTROLLER", raising=False)
    return dev, host


@pytest.mark.parametrize("name", [n for n in golden_names("trace_") if "ve_" not in n])
def test_default_argument_estimator_log_prob_on_the_device_controller(name, monkeypatch):
    """The fixtures' models and probes, `log_prob` with the reference's default arguments (dopri5, atol = rtol = 1e-4,
    min_step 1e-6): device route vs host route vs the float64 oracle.  Only the samples with linearly independent probes
    take part (the error norm is global: an ill-defined estimate would steer everyone).  (Not the VE model without sigma
    normalisation: its random-init right-hand side at t = 1e-5 drives any adaptive controller to vanishing steps.)"""
    meta, a = load_golden(name)
    so = score_oracle(meta, a, torch.float64)
    for kind, kw, probes in (("hutchpp", dict(hutchpp=True, hpp_rank=meta["hpp_rank"], hpp_vecs=meta["hpp_vecs"]), [a["S"], a["G"]]),
                             ("xtrace", dict(xtrace=True, xt_vecs=meta["xt_vecs"]), [a["O"]])):
        ok = well_posed(probes[0])
        x = a["x"][ok]
        cond = a["cond"][ok] if "cond" in a else None
        pr = [p[:, ok].contiguous() for p in probes]
        sm = score_model(meta, a, DEV, **kw)

        def run():
            q = _with_probes(monkeypatch, pr)
            lp = sm.log_prob(x.to(DEV), conditional=None if cond is None else cond.to(DEV))
            assert not q
            return lp, dict(sm.last_solver_stats)
        (ld, sd), (lh, sh) = _both_routes(monkeypatch, run)
        assert "chunks" in sd and "chunks" not in sh, (sd, sh)          # it really was the device loop / the host loop
        assert (sd["attempts"], sd["accepted"]) == (sh["attempts"], sh["accepted"]), (name, kind, sd, sh)
        assert sd["accepted"] >= 3 and sd["chunks"] <= 3
        # (the two controllers' table rows differ in the last bit -- device vs libm transcendentals; the exact trace
        # carries that to 1e-5 at most, XTrace's R^-1 amplifies it a little further)
        assert _logp_err(ld, lh.cpu()) < 5e-5, (name, kind)
        # rounds 1-3's route: host controller AND the torch statement of the estimator
        monkeypatch.setenv("FF_HOST_CONTROLLER", "1")
        monkeypatch.setenv("FF_TORCH_ESTIMATOR", "1")
        lt, stt = run()
        monkeypatch.delenv("FF_HOST_CONTROLLER")
        monkeypatch.delenv("FF_TORCH_ESTIMATOR")
        assert abs(stt["attempts"] - sd["attempts"]) <= 2 and _logp_err(ld, lt.cpu()) < ADAPT_TOL, (name, kind, sd, stt)
        ref = so.log_prob(x.double(), None if cond is None else cond.double(), "dopri5", {"min_step": 1e-6}, kind,
                          tuple(p.double() for p in pr) if kind == "hutchpp" else pr[0].double())
        assert _logp_err(ld, ref.float()) < ADAPT_TOL, (name, kind)


def test_notebook_estimator_calls_and_full_rank_sketch(monkeypatch):
    """demo_diffusion.ipynb's timed calls: a 2-d VE model with 3x128 hidden units, `hutchpp=True` / `xtrace=True` with one
    probe each and every other argument at its default.  Device route vs host route; several thousand points take a
    handful of chunks (one read-back each).  With rank = D the sketch spans everything and Hutch++ IS the exact trace."""
    sm, so32, _ = _seeded_score_model(2, 0, [128] * 3, "VESDE", False, 901)
    torch.manual_seed(3)
    x0 = torch.randn(5000, 2) * 0.5
    for kw in (dict(hutchpp=True), dict(xtrace=True)):
        sm.hutchpp, sm.xtrace = kw.get("hutchpp", False), kw.get("xtrace", False)

        def run():
            torch.manual_seed(40)
            lp = sm.log_prob(x0.to(DEV))
            return lp, dict(sm.last_solver_stats)
        (ld, sd), (lh, sh) = _both_routes(monkeypatch, run)
        assert (sd["attempts"], sd["accepted"]) == (sh["attempts"], sh["accepted"]), (kw, sd, sh)
        assert sd["chunks"] <= 4 and _logp_err(ld, lh.cpu()) < 5e-5, kw
        # the run repeats bit for bit
        assert torch.equal(run()[0], ld)
    sm.hutchpp, sm.xtrace, sm.hpp_rank = True, False, 2
    torch.manual_seed(41)
    lp_pp = sm.log_prob(x0[:800].to(DEV))
    sm.hutchpp = False
    lp_exact = sm.log_prob(x0[:800].to(DEV))
    assert max_rel(lp_pp.cpu(), lp_exact.cpu(), floor=1.0) < 2e-4
    sm.hpp_rank = 1


def test_estimator_solves_on_other_pairs_and_first_step(monkeypatch):
    """bosh3 (three evaluation rows per attempt) and `first_step` (no initial-step launches) with the estimator (one
    sketch probe: always well posed; three residual probes)."""
    sm, _, _ = _seeded_score_model(5, 3, [64, 100], "VPSDE", True, 77)
    sm.hutchpp, sm.hpp_rank, sm.hpp_vector = True, 1, 3
    torch.manual_seed(6)
    x0, cond = torch.randn(300, 5) * 0.5, torch.randn(300, 3)
    for kwargs in (dict(method="bosh3"), dict(options={"first_step": 0.01, "min_step": 1e-6})):
        def run():
            torch.manual_seed(9)
            lp = sm.log_prob(x0.to(DEV), conditional=cond.to(DEV), **kwargs)
            return lp, dict(sm.last_solver_stats)
        (ld, sd), (lh, sh) = _both_routes(monkeypatch, run)
        assert (sd["attempts"], sd["accepted"]) == (sh["attempts"], sh["accepted"]), (kwargs, sd, sh)
        assert _logp_err(ld, lh.cpu()) < 5e-5, kwargs


def test_estimator_solves_at_size_properties(monkeypatch):
    """Size-independent properties of an estimator solve at 2^16 x 16-d (config 2's network): flipping the sign of every
    probe leaves Hutch++ and XTrace unchanged BIT FOR BIT (q -> -q, u -> -u: every product keeps its value), a repeated run
    is bitwise equal, and on a fixed grid a slice of the batch solved on its own returns the rows of the whole solve."""
    from flowfusion_amd import trace_estimators as TE
    sm, _, _ = _seeded_score_model(16, 0, [256] * 4, "VPSDE", True, 12)
    B = 1 << 16
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(B, 16, device=DEV, generator=g) * 0.8
    S, G, O = (torch.sign(torch.randn(n, B, 16, device=DEV, generator=g)) for n in (1, 2, 2))
    opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / 5}
    for kind, probes in (("hutchpp", [S, G]), ("xtrace", [O])):
        sm.hutchpp, sm.xtrace, sm.hpp_rank, sm.hpp_vector, sm.xt_vector = kind == "hutchpp", kind == "xtrace", 1, 2, 2

        def run(sign, rows=slice(None), **kw):
            queue = [sign * p[:, rows].contiguous() for p in probes]
            monkeypatch.setattr(TE, "draw_probes", lambda n, like: queue.pop(0))
            return sm.log_prob(x[rows].contiguous(), **kw)
        fixed = run(1.0, method="rk4", options=opts)
        assert torch.isfinite(fixed).all() or kind == "xtrace"          # (two equal probes among 2^16 samples: XTrace divides by ~0 there)
        ok = torch.isfinite(fixed)
        assert ok.float().mean() > 0.999
        assert torch.equal(run(-1.0, method="rk4", options=opts)[ok], fixed[ok])
        assert torch.equal(run(1.0, method="rk4", options=opts)[ok], fixed[ok])
        part = run(1.0, rows=slice(1000, 3333), method="rk4", options=opts)
        assert torch.equal(part[ok[1000:3333]], fixed[1000:3333][ok[1000:3333]])
        if kind == "hutchpp":                                           # the reference's default solver: a re-run is bitwise equal
            a, b = run(1.0), run(1.0)
            assert torch.equal(a, b) and torch.isfinite(a).all() and sm.last_solver_stats["chunks"] <= 2
            assert torch.equal(run(-1.0), a)
    sm.hutchpp = sm.xtrace = False


def test_estimator_probes_from_the_counter_based_stream():
    """probe="philox" on a Hutch++ / XTrace model: S, G, O are the signs of the library's counter-based normals keyed by
    (seed, global row, a reserved index per probe) -- drawn on the device, the same whatever the sharding: a slice of the
    batch solved on its own returns the rows of the whole solve bit for bit (what distributed.log_prob_sharded relies on),
    and the signs equal the numpy restatement of the stream."""
    import numpy as np
    from flowfusion_amd import _native
    from flowfusion_amd.distributed import log_prob_sharded
    from tests._philox import normals
    sm, _, _ = _seeded_score_model(8, 2, [128, 128], "VPSDE", True, 44)
    torch.manual_seed(2)
    B = 400
    x, cond = torch.randn(B, 8, device=DEV) * 0.6, torch.randn(B, 2, device=DEV)
    opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / 6}
    lo, hi = 130, 333
    for kind in ("hutchpp", "xtrace"):
        sm.hutchpp, sm.xtrace, sm.hpp_rank, sm.hpp_vector, sm.xt_vector = kind == "hutchpp", kind == "xtrace", 2, 3, 2
        kw = dict(conditional=cond, method="rk4", options=opts, probe="philox")
        lp = sm.log_prob(x, seed=71, **kw)
        first = sm.S if kind == "hutchpp" else sm.O
        z = normals(71, 0, B, 8, [_native.TRACE_PROBE_NOISE_BASE, _native.TRACE_PROBE_NOISE_BASE + 1])
        far = np.abs(z) > 1e-5
        assert np.array_equal(first.cpu().numpy()[far], np.where(z >= 0, 1.0, -1.0)[far]) and far.mean() > 0.999
        if kind == "hutchpp":
            zg = normals(71, 0, B, 8, [_native.TRACE_PROBE_NOISE_BASE + 0x8000 + c for c in range(3)])
            farg = np.abs(zg) > 1e-5
            assert np.array_equal(sm.G.cpu().numpy()[farg], np.where(zg >= 0, 1.0, -1.0)[farg])
        ok = torch.isfinite(lp).reshape(-1)
        assert ok.float().mean() > 0.97                                 # (two equal probes of a sample: ill-posed, as in the reference)
        assert torch.equal(sm.log_prob(x, seed=71, **kw)[ok], lp[ok]) and not torch.equal(sm.log_prob(x, seed=72, **kw)[ok], lp[ok])
        part = sm.log_prob(x[lo:hi].contiguous(), seed=71, sample_offset=lo, **{**kw, "conditional": cond[lo:hi].contiguous()})
        assert torch.equal(part[ok[lo:hi]], lp[lo:hi][ok[lo:hi]])
        assert torch.equal(log_prob_sharded(sm, x, cond, seed=71, method="rk4", options=opts)[ok], lp[ok])
    sm.hutchpp = sm.xtrace = False
    with pytest.raises(ValueError, match="hutchinson=True, hutchpp=True or xtrace=True"):
        sm.log_prob(x, conditional=cond, probe="philox")
