"""GPU tier: a DETERMINISTIC guard for the kernels that share LDS between the wavefronts of a workgroup.

Round 3 found two synchronisation holes in the cooperative twin of the fused kernel (csrc/ff_mlp_ode.hpp) that were invisible
solo and only showed when another process shared the card: (1) a wavefront that started late zero-filled the shared stage
slots AFTER another had stored the caller's first stage (k1_in: every adaptive attempt) and before that one read it back;
(2) with an odd number of hidden layers the activation-exchange buffer count restarted with every evaluation, so the last
exchange of one evaluation and the first of the next used the same buffer with a single barrier between a slow wavefront's
reads and a fast one's stores.  Their regression test puts three processes on the card and compares fingerprints -- a
regression can pass it.

Here the skew is built in (csrc/ff_skew.h, flowfusion_amd/build.py VARIANTS: test-only libraries, never the product): one
wavefront of every workgroup is held back with s_sleep exactly where a late wavefront does damage if a barrier is missing.
Single process, nobody else on the card needed:
  * `skew_unfix` = the skewed twin with round 3's two fixes REMOVED: its results must be WRONG (else this file guards nothing);
  * `skew` = HEAD's kernels under the same skew: bitwise the one-wavefront kernel's results -- the twin in state-only and
    tangent-column form, an adaptive attempt (k1_in) and an odd hidden-layer count; the WIDE catch-all and one
    split-precision kernel (their own LDS sharing: one exchange buffer and two barriers per layer; a ring of weight granules
    with one barrier each) bitwise against their un-skewed product builds.
"""
import ctypes
import os

import pytest
import torch

from flowfusion_amd import _native
from tests.test_gpu_parity import DEV, _seeded_score_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def libs(built_library):
    assert torch.cuda.is_available(), "the gpu tier needs a GPU"
    from flowfusion_amd import build
    if not all(build.variant_lib(v).exists() for v in ("skew", "skew_unfix")):
        build.build()                                     # (normally built by __graft_entry__.build() with the product)
    return {"product": built_library, "skew": _native.load_library(build.variant_lib("skew")),
            "unfix": _native.load_library(build.variant_lib("skew_unfix"))}


def _plan(L, net, mode):
    """The plan `L` makes for the network of `net` -- kernel ids are per library, the packed layout must be the product's."""
    ref = net.plan(mode)
    hidden = [int(l.out_features) for l in net.linears[:-1]]
    p = _native.PlanStruct()
    arr = (ctypes.c_int * len(hidden))(*hidden)
    prm = (ctypes.c_float * 2)(0.0, 0.0)
    rc = L.ff_mlp_plan_prec(net.dim, net.cond_dim, len(hidden), arr, mode, _native.ACT_SILU, prm, int(ref.precision), ctypes.byref(p))
    assert rc == 0, rc
    for f in ("dim", "cond_dim", "n_hidden", "width", "dregs", "cregs", "tile", "precision"):
        assert getattr(p, f) == getattr(ref, f), f
    return p


def _launch(L, sm, mode, x, table, n_evals, k1=None, n_aux=0, probe=None, coop=None):
    """One raw ff_mlp_ode_launch through library `L`; returns (x_out, aux_0 or None, dlogp or None)."""
    net = sm._net()
    plan = _plan(L, net, mode)
    wpack = net.wpack(x.device, mode)
    B, D = x.shape
    out = torch.full_like(x, -123.0)
    aux = torch.full_like(x, -321.0) if n_aux else None
    lp = torch.full((B,), -55.0, device=x.device) if mode else None
    a = _native.OdeArgs()
    a.x_in, a.x_out, a.wpack, a.etab = x.data_ptr(), out.data_ptr(), wpack.data_ptr(), table.data_ptr()
    a.batch, a.n_evals, a.mode = B, n_evals, mode
    a.stage_slots = 0
    if k1 is not None:
        a.k1_in = k1.data_ptr()
    if n_aux:
        a.n_aux = 1
        a.aux_out[0] = aux.data_ptr()
    if mode:
        a.dlogp_out = lp.data_ptr()
        a.probe = probe.data_ptr()
    prev = os.environ.pop("FF_COOP", None)
    if coop is not None:
        os.environ["FF_COOP"] = "1" if coop else "0"
    try:
        rc = L.ff_mlp_ode_launch(ctypes.byref(plan), ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    finally:
        os.environ.pop("FF_COOP", None)
        if prev is not None:
            os.environ["FF_COOP"] = prev
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out, aux, lp


def _attempt_table(sm, n_rows, mode, dev):
    """Rows as an adaptive attempt has them: stage slot 0 comes in through k1_in, row i fills slot i + 1 from
    y + 0.1 k[0] + 0.05 k[i]; aux_0 = y + 0.2 sum_s k[s]."""
    net = sm._net()
    width = net.width(mode)
    t = torch.linspace(0.2, 0.8, n_rows)
    a, b, c1, _ = sm._schedule(t, "ode")
    rows = torch.zeros(n_rows + 2, 32 + width)
    rows[:n_rows, 0], rows[:n_rows, 1] = a, b
    ints = rows.view(torch.int32)
    for i in range(n_rows):
        ints[i, 4] = i + 1
        rows[i, 8] = 0.1
        if i:
            rows[i, 8 + i] = 0.05
    rows[:n_rows, 32:32 + c1.shape[1]] = c1
    rows[n_rows, 8:8 + n_rows + 1] = 0.2
    ints[n_rows, 3] = 1                                  # use_y of aux_0
    return rows.to(dev)


def test_unfixed_twin_fails_under_skew_and_head_does_not(libs):
    torch.manual_seed(5)
    # (1) the late zero fill: an adaptive attempt (k1_in) on an EVEN number of hidden layers (the second hole cannot fire)
    sm, _, _ = _seeded_score_model(16, 0, [256] * 4, "VPSDE", True, 21)
    B = 200                                              # 13 tiles: the twin's regime
    x = torch.randn(B, 16, device=DEV)
    k1 = torch.randn(B, 16, device=DEV)
    tab = _attempt_table(sm, 5, 0, DEV)
    ref_x, ref_aux, _ = _launch(libs["product"], sm, 0, x, tab, 5, k1=k1, n_aux=1, coop=False)      # one-wavefront kernel
    twin_x, twin_aux, _ = _launch(libs["product"], sm, 0, x, tab, 5, k1=k1, n_aux=1, coop=True)
    assert torch.equal(twin_aux, ref_aux) and torch.equal(twin_x, ref_x)
    assert (ref_aux - x).abs().max() > 1e-3                                                        # the attempt did something
    _, bad_aux, _ = _launch(libs["unfix"], sm, 0, x, tab, 5, k1=k1, n_aux=1, coop=True)
    assert not torch.equal(bad_aux, ref_aux), "the skewed kernel WITHOUT the barrier behind the zero fill must lose k1_in"
    for _ in range(3):
        got_x, got_aux, _ = _launch(libs["skew"], sm, 0, x, tab, 5, k1=k1, n_aux=1, coop=True)
        assert torch.equal(got_aux, ref_aux) and torch.equal(got_x, ref_x)
    # ... the same with tangent columns (Hutchinson): the divergence slots are per lane, the state slots shared
    probe = torch.sign(torch.randn(B, 16, device=DEV))
    tab1 = _attempt_table(sm, 5, 1, DEV)
    r = _launch(libs["product"], sm, 1, x, tab1, 5, k1=k1, n_aux=1, probe=probe, coop=False)
    b = _launch(libs["unfix"], sm, 1, x, tab1, 5, k1=k1, n_aux=1, probe=probe, coop=True)
    g = _launch(libs["skew"], sm, 1, x, tab1, 5, k1=k1, n_aux=1, probe=probe, coop=True)
    assert not torch.equal(b[1], r[1])
    assert all(torch.equal(u, v) for u, v in zip(g, r))

    # (2) exchange buffers: an ODD number of hidden layers, a plain fixed-grid table (no k1_in: the first hole cannot fire)
    sm3, _, _ = _seeded_score_model(16, 0, [256] * 3, "VPSDE", True, 22)
    opts = {"step_size": (1.0 - float(sm3.sde.epsilon)) / 3}
    t_span = torch.tensor([1.0, float(sm3.sde.epsilon)])
    for mode in (0, 1):
        tab = sm3._ode_table(t_span, "rk4", opts, mode).to(DEV)
        n = tab.shape[0]
        kw = dict(probe=probe) if mode else {}
        r = _launch(libs["product"], sm3, mode, x, tab, n, coop=False, **kw)
        t = _launch(libs["product"], sm3, mode, x, tab, n, coop=True, **kw)
        assert torch.equal(t[0], r[0]) and (mode == 0 or torch.equal(t[2], r[2]))
        b = _launch(libs["unfix"], sm3, mode, x, tab, n, coop=True, **kw)
        assert not torch.equal(b[0], r[0]), "the skewed kernel with per-evaluation buffer counts must read overwritten activations"
        for _ in range(3):
            g = _launch(libs["skew"], sm3, mode, x, tab, n, coop=True, **kw)
            assert torch.equal(g[0], r[0]) and (mode == 0 or torch.equal(g[2], r[2]))
    # an even count under the un-fixed build WITHOUT k1_in: neither hole can fire -- the skew alone changes nothing
    tab = sm._ode_table(t_span, "rk4", opts, 0).to(DEV)
    assert torch.equal(_launch(libs["unfix"], sm, 0, x, tab, tab.shape[0], coop=True)[0],
                       _launch(libs["product"], sm, 0, x, tab, tab.shape[0], coop=False)[0])


def test_wide_catch_all_and_split_kernel_under_skew(libs):
    """The other kernels that share LDS across wavefronts, held back the same way: bitwise their un-skewed product builds."""
    torch.manual_seed(6)
    # WIDE: one exchange buffer, two barriers per layer, operands read from LDS during the next layer.  The skew library
    # holds the wide kernel at width 256 / 32 dimensions; the product serves that shape with a ONE-WAVEFRONT kernel (same packed
    # layout, same FMA chains): the skewed wide kernel must reproduce it bit for bit.
    sw, _, _ = _seeded_score_model(20, 0, [200, 256, 256], "VESDE", False, 23)
    assert "wide" not in _native.kernel_name(sw._net().plan(0))
    assert b"wide" in libs["skew"].ff_plan_kernel_name(ctypes.byref(_plan(libs["skew"], sw._net(), 0)))
    x = torch.randn(70, 20, device=DEV)
    t_span = torch.tensor([1.0, float(sw.sde.epsilon)])
    tab = sw._ode_table(t_span, "rk4", {"step_size": 0.34}, 0).to(DEV)
    k1 = torch.randn(70, 20, device=DEV)
    ref = _launch(libs["product"], sw, 0, x, tab, tab.shape[0], coop=False)
    got = _launch(libs["skew"], sw, 0, x, tab, tab.shape[0])
    assert torch.equal(got[0], ref[0]) and torch.isfinite(ref[0]).all()
    att = _attempt_table(sw, 4, 0, DEV)
    ref = _launch(libs["product"], sw, 0, x, att, 4, k1=k1, n_aux=1, coop=False)
    got = _launch(libs["skew"], sw, 0, x, att, 4, k1=k1, n_aux=1)
    assert torch.equal(got[1], ref[1])
    # split precision (bf16x2, three hidden layers of 128): a ring of weight granules, one barrier per granule
    ss, _, _ = _seeded_score_model(8, 0, [128] * 3, "VESDE", False, 24)
    ss.precision = "bf16x2"
    assert "split2_h128_n3_t0" in _native.kernel_name(ss._net().plan(0))
    x = torch.randn(300, 8, device=DEV)
    tab = ss._ode_table(torch.tensor([1.0, float(ss.sde.epsilon)]), "rk4", {"step_size": 0.26}, 0).to(DEV)
    ref = _launch(libs["product"], ss, 0, x, tab, tab.shape[0])
    got = _launch(libs["skew"], ss, 0, x, tab, tab.shape[0])
    assert torch.equal(got[0], ref[0]) and torch.isfinite(ref[0]).all()
