"""precision="bf16x3" / "bf16x2": the opt-in split-precision kernels (csrc/ff_mlp_ode_split.hpp, FF_PREC_BF16X3 / _BF16X2).

CPU tier: plan selection, the host packer (the fragment stream decodes back to the fp32 weights bit for bit:
hi + mid + lo is an exact three-way split), error behaviour.
GPU tier: the same parity battery the f32 path passes -- golden hybrids, oracle fp32 / fp64 on seeded inputs
(BASELINE configs 2 and 3 included), ragged batches, Hutchinson tangents, conditional inputs, flows -- at the SAME
tolerances (2e-5; north_star's bar for log_prob is 1e-4): six bf16 products per term carry ~24 significand bits, so
the split path sits at fp32 rounding level, not at bf16 level (a single-product bf16 path would miss by 1e-2).
"bf16x2" (two round-to-nearest parts, three products: operands to 16 bits, unbiased) runs the same battery at the same
tolerances: its per-layer error is ~20x an fp32 dot product's and still two orders below the bar.
"""
import ctypes

import numpy as np
import pytest
import torch

from flowfusion_amd import _native
from flowfusion_amd import diffusion as D
from flowfusion_amd import flow as F
from flowfusion_amd.fused import MODE_EXACT, MODE_HUTCH, MODE_STATE
from tests._util import flow_model, flow_oracle, golden_names, load_golden, max_rel, score_model, score_oracle

DEV = "cuda"
STATE_TOL = 2e-5
LOGP_TOL = 2e-5
PRECS = ["bf16x3", "bf16x2"]
# log-densities (Hutchinson column pairs, exact trace) under precision=: the two-part kernels.  Round 3 froze the three-part
# option at its state-only kernels (the second record of the headline workload): its divergence modes raise.
LOGP_PRECS = ["bf16x2"]


def _logp_raises(fn):
    with pytest.raises(NotImplementedError, match="bf16x3"):
        fn()


def _kidx(s, q, j):
    return 32 * s + 16 * (j >> 2) + 4 * q + (j & 3)


def _decode_group(words, g):
    """[64 lanes, 8 elements] fp32 of group g of the stream: hi + mid + lo (lane = 16 quad + row of the 16-row tile)."""
    frag = words[g * 768:(g + 1) * 768].reshape(3, 64, 4)                       # part, lane, word
    halves = np.stack([frag & 0xFFFF, frag >> 16], axis=-1).reshape(3, 64, 8)   # element j = 2 word + half
    vals = (halves.astype(np.uint32) << 16).view(np.float32)
    tot = vals[0].astype(np.float64) + vals[1].astype(np.float64) + vals[2].astype(np.float64)
    return tot.astype(np.float32)                                               # [lane, j]


def test_plan_and_packer_roundtrip(built_library):
    torch.manual_seed(3)
    sm = D.ScoreModel(D.MLP(11, 5, 8, [200, 256, 77]), D.VPSDE(), no_sigma=True, precision="bf16x3").eval()
    net = sm._net()
    plan = net.plan(MODE_STATE)
    assert (plan.precision, plan.tile, plan.width, plan.dregs, plan.cregs, plan.n_hidden) == (1, 32, 256, 8, 8, 3)
    assert _native.kernel_name(plan) == "mlp_ode_split_h256_n3_t0"
    assert _native.samples_per_workgroup(plan, MODE_STATE) == 128
    _logp_raises(lambda: net.plan(MODE_HUTCH))                       # three parts: state-only kernels since round 3
    sm2 = D.ScoreModel(sm.model, D.VPSDE(), no_sigma=True, precision="bf16x2").eval()
    n2 = sm2._net()
    assert _native.kernel_name(n2.plan(MODE_HUTCH)) == "mlp_ode_split2_h256_n3_t1"
    assert _native.samples_per_workgroup(n2.plan(MODE_HUTCH), MODE_HUTCH) == 64
    assert n2.wpack("cpu", MODE_HUTCH) is n2.wpack("cpu", MODE_STATE)      # one layout for both instantiations
    pack = net.wpack("cpu", MODE_STATE)
    words = pack.numpy().view(np.uint32)
    NR, NS, NH, H = 16, 8, 3, 256
    n_gran = (NR + (NH - 1) * NR * NS + NS) // 8
    assert words.size == n_gran * 6144 + (NH - 1) * H + 16
    Ws = [l.weight.detach().numpy() for l in sm.model.NN]
    bs = [l.bias.detach().numpy() for l in sm.model.NN]
    E = 8                                                            # time-embedding columns of the first layer (sin, cos of 4 frequencies)
    g = 0
    # layer 1: one k-step, features 0..15 = state columns, 16..31 = conditional columns
    for rt in range(NR):
        got = _decode_group(words, g)
        g += 1
        for lane in (0, 15, 16, 37, 63):
            row, q = 16 * rt + (lane & 15), lane >> 4
            for j in range(8):
                f = _kidx(0, q, j)
                col = (E + f if f < 11 else None) if f < 16 else (E + 11 + f - 16 if f - 16 < 5 else None)
                exp = Ws[0][row, col] if (row < 200 and col is not None) else 0.0
                assert got[lane, j] == np.float32(exp)
    # hidden -> hidden layers, k-major: exact reconstruction of the whole (zero-padded) matrix
    widths = [200, 256, 77]
    for l in range(1, NH):
        rec = np.zeros((H, H), np.float32)
        for s_ in range(NS):
            for rt in range(NR):
                got = _decode_group(words, g)
                g += 1
                for lane in range(64):
                    for j in range(8):
                        rec[16 * rt + (lane & 15), _kidx(s_, lane >> 4, j)] = got[lane, j]
        exp = np.zeros((H, H), np.float32)
        exp[:widths[l], :widths[l - 1]] = Ws[l]
        assert np.array_equal(rec, exp)                              # hi + mid + lo == the fp32 weight, bit for bit
    rec = np.zeros((16, H), np.float32)
    for s_ in range(NS):
        got = _decode_group(words, g)
        g += 1
        for lane in range(64):
            for j in range(8):
                rec[lane & 15, _kidx(s_, lane >> 4, j)] = got[lane, j]
    exp = np.zeros((16, H), np.float32)
    exp[:11, :77] = Ws[NH]
    assert np.array_equal(rec, exp) and g == n_gran * 8
    tail = pack.numpy()[n_gran * 6144:]
    assert np.array_equal(tail[:256], bs[1]) and np.array_equal(tail[256:256 + 77], bs[2][:77]) and not tail[256 + 77:512].any()
    assert np.array_equal(tail[512:512 + 11], bs[3]) and not tail[512 + 11:].any()


def test_two_part_packer_rounds_to_nearest(built_library):
    """FF_PREC_BF16X2: groups of two fragments; hi = bf16(w) and mid = bf16(w - hi), both round-to-nearest-even: the pair
    carries the weight to 16 significand bits (|error| <= 2^-17 |w|), the biases stay fp32."""
    torch.manual_seed(4)
    sm = D.ScoreModel(D.MLP(7, 2, 8, [256, 128]), D.VPSDE(), no_sigma=True, precision="bf16x2").eval()
    net = sm._net()
    plan = net.plan(MODE_STATE)
    assert (plan.precision, plan.tile, plan.width, plan.n_hidden) == (2, 32, 256, 2)
    assert _native.kernel_name(plan) == "mlp_ode_split2_h256_n2_t0" and _native.kernel_name(net.plan(MODE_HUTCH)) == "mlp_ode_split2_h256_n2_t1"
    words = net.wpack("cpu", MODE_STATE).numpy().view(np.uint32)
    n_gran = (16 + 128 + 8) // 8
    assert words.size == n_gran * 4096 + 256 + 16
    W1 = sm.model.NN[1].weight.detach()                                  # [128, 256]
    hi = W1.bfloat16().float()
    mid = (W1 - hi).bfloat16().float()
    for s_, rt, lane in ((0, 0, 0), (3, 5, 17), (7, 7, 63), (5, 2, 40)):
        g = 16 + s_ * 16 + rt
        frag = words[g * 512:(g + 1) * 512].reshape(2, 64, 4)
        halves = np.stack([frag & 0xFFFF, frag >> 16], axis=-1).reshape(2, 64, 8)
        vals = (halves.astype(np.uint32) << 16).view(np.float32)
        row = 16 * rt + (lane & 15)
        for j in range(8):
            k = _kidx(s_, lane >> 4, j)
            assert vals[0, lane, j] == hi[row, k].item() and vals[1, lane, j] == mid[row, k].item()
            assert abs(float(vals[0, lane, j]) + float(vals[1, lane, j]) - W1[row, k].item()) <= 2.0 ** -17 * abs(W1[row, k].item())
    tail = net.wpack("cpu", MODE_STATE).numpy()[n_gran * 4096:]
    assert np.array_equal(tail[:128], sm.model.NN[1].bias.detach().numpy()) and np.array_equal(tail[256:256 + 7], sm.model.NN[2].bias.detach().numpy())


def test_packer_two_state_tiles(built_library):
    """FF_PREC_BF16X2, 20 state dimensions, 3 conditional inputs, hidden widths (100, 128): the two-tile layout (256 wide
    on chip since round 3 froze the family: the 128-wide two-tile instances are gone) -- layer 1 = two k-steps of 16 row
    tiles (features 0..31 the state, then 0..15 the conditional inputs); hidden layer k-major over 8 k-steps; output
    layer two row tiles per k-step; biases behind the stream.  State-only kernels serve these states (config 5 is a
    sampler); the divergence modes raise."""
    torch.manual_seed(6)
    sm = D.ScoreModel(D.MLP(20, 3, 8, [100, 128]), D.VPSDE(), no_sigma=True, precision="bf16x2").eval()
    net = sm._net()
    plan = net.plan(MODE_STATE)
    assert (plan.precision, plan.width, plan.dregs, plan.n_hidden) == (2, 256, 16, 2) and net.stage_slots(MODE_STATE) == 4
    assert _native.kernel_name(plan) == "mlp_ode_split2_h256_d2_n2_t0"
    for mode in (MODE_HUTCH, MODE_EXACT):
        with pytest.raises(NotImplementedError, match="bf16x2"):
            net.plan(mode)
    words = net.wpack("cpu", MODE_STATE).numpy().view(np.uint32)
    NR, NS, H = 16, 8, 256
    n_groups = 2 * NR + NR * NS + 2 * NS
    assert n_groups % 8 == 0 and words.size == n_groups * 512 + 1 * H + 32
    E = 8                                                            # time-embedding columns in front of the state columns

    def group(g):
        frag = words[g * 512:(g + 1) * 512].reshape(2, 64, 4)
        halves = np.stack([frag & 0xFFFF, frag >> 16], axis=-1).reshape(2, 64, 8)
        return (halves.astype(np.uint32) << 16).view(np.float32)    # [part, lane, j]

    def parts(w):
        hi = torch.tensor(w).bfloat16().float().item()
        return hi, torch.tensor(w - hi).bfloat16().float().item()
    W0, W1, W2 = (l.weight.detach().numpy() for l in sm.model.NN)
    for s1, rt, lane in ((0, 0, 0), (0, 6, 21), (1, 3, 47), (1, 7, 63), (1, 12, 9)):
        vals = group(s1 * NR + rt)
        row, q = 16 * rt + (lane & 15), lane >> 4
        for j in range(8):
            f = _kidx(0, q, j)
            col = (E + f if f < 20 else None) if s1 == 0 else (E + 20 + f if f < 3 else None)
            exp = parts(float(W0[row, col])) if (row < 100 and col is not None) else (0.0, 0.0)
            assert (vals[0, lane, j], vals[1, lane, j]) == exp, (s1, rt, lane, j)
    for s_, rt, lane in ((0, 0, 5), (2, 7, 33), (3, 4, 60), (7, 15, 1)):
        vals = group(2 * NR + s_ * NR + rt)
        row, q = 16 * rt + (lane & 15), lane >> 4
        for j in range(8):
            k = _kidx(s_, q, j)
            exp = parts(float(W1[row, k])) if (k < 100 and row < 128) else (0.0, 0.0)
            assert (vals[0, lane, j], vals[1, lane, j]) == exp
    for s_, t, lane in ((0, 0, 3), (1, 1, 2), (3, 1, 20), (2, 0, 50)):
        vals = group(2 * NR + NR * NS + s_ * 2 + t)
        row, q = 16 * t + (lane & 15), lane >> 4
        for j in range(8):
            kk = _kidx(s_, q, j)
            exp = parts(float(W2[row, kk])) if (row < 20 and kk < 128) else (0.0, 0.0)
            assert (vals[0, lane, j], vals[1, lane, j]) == exp
    tail = net.wpack("cpu", MODE_STATE).numpy()[n_groups * 512:]
    assert np.array_equal(tail[:128], sm.model.NN[1].bias.detach().numpy()) and not tail[128:256].any()
    assert np.array_equal(tail[256:276], sm.model.NN[2].bias.detach().numpy()) and not tail[276:].any()


def test_split_precision_scope_and_errors(built_library):
    mk = lambda **kw: _native.make_plan(kw.get("dim", 16), kw.get("cond", 0), kw.get("hidden", [256] * 4), kw.get("mode", MODE_STATE),
                                        kw.get("act", (_native.ACT_SILU, 0.0, 0.0)), _native.PREC_BF16X3)
    assert mk().precision == 1
    px = _native.make_plan(16, 0, [256] * 4, MODE_EXACT, (_native.ACT_SILU, 0.0, 0.0), _native.PREC_BF16X2)
    assert (px.tile, _native.kernel_name(px)) == (16, "mlp_ode_split2_h256_n4_t2") and _native.samples_per_workgroup(px, MODE_EXACT) == 8
    for bad in (dict(dim=17), dict(cond=17), dict(hidden=[300, 300]), dict(hidden=[64] * 5), dict(mode=MODE_EXACT), dict(mode=MODE_HUTCH),
                dict(act=(_native.ACT_TANH, 0.0, 0.0))):
        with pytest.raises(NotImplementedError, match="bf16x3"):
            mk(**bad)
    with pytest.raises(ValueError, match="precision"):
        D.ScoreModel(D.MLP(4, 0, 8, [64]), D.VPSDE(), precision="fp8")._net()
    # a bf16x3 plan is refused by entry points it does not serve, before any HIP call
    p = mk()
    a = _native.OdeArgs()
    assert built_library.ff_mlp_ode_launch(ctypes.byref(p), ctypes.byref(a), None) == _native.FF_ERR_BADARG
    # precision is part of the cache key: flipping the attribute rebuilds the kernel-side view
    sm = D.ScoreModel(D.MLP(4, 0, 8, [64, 64]), D.VPSDE(), no_sigma=True).eval()
    n32 = sm._net()
    sm.precision = "bf16x3"
    n16 = sm._net()
    assert n16 is not n32 and n16.plan(MODE_STATE).precision == 1 and n32.plan(MODE_STATE).precision == 0


# ---- GPU tier ------------------------------------------------------------------------------------------------------
def _state_err(got, exp):
    return max_rel(got.cpu(), exp, floor=exp.abs().max().item())


def _logp_err(got, exp):
    return max_rel(got.cpu(), exp, floor=1.0)


def _seeded(Dm, C, units, sde_name, no_sigma, seed, prec="bf16x3"):
    torch.manual_seed(seed)
    m = D.MLP(n_dimensions=Dm, n_conditionals=C, embedding_dimensions=8, units=units)
    sm = D.ScoreModel(m, getattr(D, sde_name)(), no_sigma=no_sigma, precision=prec).eval()
    meta = dict(D=Dm, C=C, E=8, units=units, sde=sde_name, sde_kw={}, no_sigma=no_sigma)
    arrays = {k: v.detach().clone() for k, v in sm.state_dict().items()}
    return sm.to(DEV), score_oracle(meta, arrays), score_oracle(meta, arrays, torch.float64)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", [n for n in golden_names("hybrid_score_") if "cond32" not in n])
def test_split_against_golden_hybrids(name, prec, built_library):
    meta, a = load_golden(name)
    if meta["D"] > 16 or meta["C"] > 16:
        pytest.skip("outside the split family's shape envelope")
    sm = score_model(meta, a, DEV)
    sm.precision = prec
    cond = a.get("cond")
    cond_d = None if cond is None else cond.to(DEV)
    for run in meta["runs"]:
        m, opts = run["method"], {"step_size": run["step_size"]}
        x0, _ = sm.sample_ode_from_base(a["base"].to(DEV), conditional=cond_d, method=m, options=opts)
        assert _state_err(x0, a[f"sample_{m}"]) < STATE_TOL, (name, m)
        if prec not in LOGP_PRECS:
            continue
        sm.hutch = True
        xd = a[f"x_data_{m}"].to(DEV)
        tab = sm._ode_table(torch.tensor([float(sm.sde.epsilon), 1.0]), m, opts, 1)
        xT, dlp, _ = sm._net().integrate(xd, tab, 1, cond=cond_d, probe=a[f"e_{m}"].to(DEV))
        lp = dlp.view(-1, 1) + sm.sde.prior(xT.shape).log_prob(xT).sum(1, keepdim=True)
        assert _logp_err(lp, a[f"lp_hutch_{m}"]) < LOGP_TOL, (name, m)
        assert _state_err(xT, a[f"xT_{m}"]) < STATE_TOL
        sm.hutch = False
    assert _native.kernel_name(sm._net().plan(0)).startswith("mlp_ode_split")


CONFIGS = {
    "c2_16d_vp_4x256_rk4_100": (16, 0, [256] * 4, "VPSDE", True, "rk4", 100, 777),
    "c1_2d_ve_3x128_euler50": (2, 0, [128] * 3, "VESDE", False, "euler", 50, 1000),
    "cond_5d_c3_ragged_subvp": (5, 3, [64, 100], "SUBVPSDE", False, "midpoint", 30, 129),
    "cond_16d_c16_dopri5_fixed": (16, 16, [256, 256], "VPSDE", True, "dopri5_fixed", 20, 200),
    "one_hidden_layer": (8, 0, [200], "VESDE", False, "heun3", 25, 333),
    "four_ragged_hidden_layers": (12, 2, [96, 64, 128, 96], "VPSDE", False, "rk4_classic", 20, 150),
}


@pytest.mark.gpu
@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", list(CONFIGS))
def test_split_sampling_and_log_prob_against_oracle(name, prec, built_library):
    Dm, C, units, sde_name, no_sigma, method, nsteps, B = CONFIGS[name]
    sm, so32, so64 = _seeded(Dm, C, units, sde_name, no_sigma, 11, prec)
    torch.manual_seed(1234)
    base = torch.randn(B, Dm)
    cond = torch.randn(B, C) if C else None
    cd = None if cond is None else cond.to(DEV)
    opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / nsteps}
    got, _ = sm.sample_ode_from_base(base.to(DEV), conditional=cd, method=method, options=opts)
    ref64 = so64.sample_ode_from_base(base.double(), None if cond is None else cond.double(), method, opts).float()
    e_gpu, e_cpu = _state_err(got, ref64), _state_err(so32.sample_ode_from_base(base, cond, method, opts), ref64)
    assert e_gpu < STATE_TOL, (name, e_gpu, e_cpu)
    print(f"\n[{prec}] {name}: state error vs float64 {e_gpu:.2e} (fp32 CPU oracle {e_cpu:.2e})")
    # Hutchinson log-density (value / tangent column pairs): the reference's CPU-drawn probe
    sm.hutch = True
    Bl = min(B, 256)
    x0 = torch.randn(Bl, Dm) * 0.8 + 0.3
    torch.manual_seed(99)
    if prec not in LOGP_PRECS:
        return _logp_raises(lambda: sm.log_prob(x0.to(DEV), conditional=None if cd is None else cd[:Bl], method=method, options=opts))
    lp = sm.log_prob(x0.to(DEV), conditional=None if cd is None else cd[:Bl], method=method, options=opts)
    e = sm.e.cpu()
    ref64 = so64.log_prob(x0.double(), None if cond is None else cond[:Bl].double(), method, opts, "hutch", e.double()).float()
    assert lp.shape == (Bl, 1)
    assert _logp_err(lp, ref64) < LOGP_TOL, name
    # and against the f32 kernels on the same inputs: two fp32-class evaluations of the same thing
    sm.precision = "f32"
    torch.manual_seed(99)
    lp32 = sm.log_prob(x0.to(DEV), conditional=None if cd is None else cd[:Bl], method=method, options=opts)
    assert _logp_err(lp, lp32.cpu()) < LOGP_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("prec", PRECS)
def test_split_adaptive_default_calls(prec, built_library):
    """The reference's default arguments (adaptive dopri5; diffusion.py:566-640, 756-815) on the split kernels: one launch per
    attempted step with the FSAL stage preloaded and the new state / last stage / midpoint / error estimate leaving as
    combinations of the stage slots.  Against the oracle's dopri5 (two adaptive solves: 2e-4), and the step sequence of
    the f32 kernels (same controller, fp32-class right-hand sides: same number of attempts)."""
    sm, so32, so64 = _seeded(16, 3, [256, 256, 200], "VESDE", False, 23, prec)
    torch.manual_seed(5)
    base, cond = torch.randn(300, 16), torch.randn(300, 3)
    x0, _ = sm.sample_ode_from_base(base.to(DEV), conditional=cond.to(DEV))
    stats = dict(sm.last_solver_stats)
    ref = so64.sample_ode_from_base(base.double(), cond.double(), "dopri5", None, atol=1e-4, rtol=1e-4).float()
    assert _state_err(x0, ref) < 2e-4
    if prec in LOGP_PRECS:
        sm.hutch = True
        xd = torch.randn(64, 16) * 3
        torch.manual_seed(9)
        lp = sm.log_prob(xd.to(DEV), conditional=cond[:64].to(DEV))
        e = sm.e.cpu()
        ref = so64.log_prob(xd.double(), cond[:64].double(), "dopri5", {"min_step": 1e-6}, "hutch", e.double(), atol=1e-4, rtol=1e-4).float()
        assert _logp_err(lp, ref) < 2e-4
        sm.hutch = False
    sm.precision = "f32"
    y0, _ = sm.sample_ode_from_base(base.to(DEV), conditional=cond.to(DEV))
    # (the same controller on fp32-class right-hand sides: a similar step sequence, see test_split_exact_trace)
    assert abs(sm.last_solver_stats["attempts"] - stats["attempts"]) <= 3 and _state_err(x0, y0.cpu()) < 5e-4


@pytest.mark.gpu
@pytest.mark.parametrize("prec", PRECS)
def test_split_euler_maruyama(prec, built_library):
    """sample_sde under precision=: noise rows on the split kernels.  (1) the reference's loop with an injected normal
    stream against the oracle (diffusion.py:510-563 restated), conditional VE model, 60 steps; (2) in-kernel counter-based
    noise: equals the same kernel fed tests/_philox.py's numbers through the noise buffer, and is bitwise independent
    of how the batch is cut into launches; (3) the public method runs and returns finite means."""
    from tests._philox import normals
    sm, so32, _ = _seeded(12, 4, [128, 256], "VESDE", False, 31, prec)
    B, steps = 200, 60
    torch.manual_seed(2)
    prior = torch.randn(B, 12) * float(sm.sde.sigma_max)
    cond = torch.randn(B, 4)
    draws = [torch.randn(B, 12) for _ in range(steps)]
    it = iter(draws)
    got = sm._sample_sde_from(prior.to(DEV), lambda like: next(it).to(DEV), cond.to(DEV), steps)
    ref = so32.sample_sde(prior, draws, cond, steps=steps)
    assert _state_err(got, ref) < STATE_TOL
    seed, off = 24680, 7000
    a = sm._sample_sde_from(prior.to(DEV), None, cond.to(DEV), 12, rng=(seed, off))
    z = torch.from_numpy(normals(seed, off, B, 12, list(range(12))))
    it = iter(z)
    b = sm._sample_sde_from(prior.to(DEV), lambda like: next(it).to(DEV), cond.to(DEV), 12)
    assert _state_err(a, b.cpu()) < STATE_TOL
    parts = [sm._sample_sde_from(prior[lo:hi].to(DEV), None, cond[lo:hi].to(DEV), 12, rng=(seed, off + lo))
             for lo, hi in ((0, 37), (37, 165), (165, B))]
    assert torch.equal(torch.cat(parts), a)
    torch.manual_seed(4)
    out = sm.sample_sde((64, 12), conditional=cond[:64].to(DEV), steps=20)
    assert out.shape == (64, 12) and torch.isfinite(out).all()
    assert _native.kernel_name(sm._net().plan(0)).startswith("mlp_ode_split")


EXACT = {
    "d2_ve_notebook": (2, 0, [128] * 3, "VESDE", False, "rk4", 20, 200),            # 3 columns per sample: 5 samples per block
    "d7_c3_subvp": (7, 3, [200, 64], "SUBVPSDE", False, "heun3", 15, 77),           # 8 columns: 2 samples per block
    "d16_vp_4x256": (16, 0, [256] * 4, "VPSDE", True, "rk4", 25, 100),              # 17 columns: passes of 15 + 1 tangents
    "d15_c16_one_layer": (15, 16, [256], "VPSDE", True, "euler", 30, 45),           # 16 columns: one sample per block
}


@pytest.mark.gpu
@pytest.mark.parametrize("prec", LOGP_PRECS)
@pytest.mark.parametrize("name", list(EXACT))
def test_split_exact_trace(name, prec, built_library):
    """The reference's default divergence (exact trace, diffusion.py:483-503) on the split kernels: a value column and its
    unit-tangent columns share a column block of 16, the slope travels through the LDS crossbar; fixed grid against the
    float64 oracle's autograd trace, and the reference's DEFAULT call (adaptive dopri5 + exact trace) against the f32
    kernels' (same accept / reject sequence)."""
    Dm, C, units, sde_name, no_sigma, method, nsteps, B = EXACT[name]
    sm, so32, so64 = _seeded(Dm, C, units, sde_name, no_sigma, 57, prec)
    assert _native.kernel_name(sm._net().plan(MODE_EXACT)).endswith("_t2")
    torch.manual_seed(8)
    x0 = torch.randn(B, Dm) * 0.7
    cond = torch.randn(B, C) if C else None
    cd = None if cond is None else cond.to(DEV)
    opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / nsteps}
    lp = sm.log_prob(x0.to(DEV), conditional=cd, method=method, options=opts)
    ref = so64.log_prob(x0.double(), None if cond is None else cond.double(), method, opts, "exact").float()
    assert lp.shape == (B, 1) and _logp_err(lp, ref) < LOGP_TOL, name
    lpd = sm.log_prob(x0[:40].to(DEV), conditional=None if cd is None else cd[:40])      # dopri5, exact trace, min_step 1e-6
    stats = dict(sm.last_solver_stats)
    sm.precision = "f32"
    lp32 = sm.log_prob(x0[:40].to(DEV), conditional=None if cd is None else cd[:40])
    # The same controller on fp32-class right-hand sides walks a similar step sequence -- not the same one: the error ratio
    # feeds back into the step size (err ~ dt^5), so a 1e-7 difference in the right-hand side grows to percents of the ratio
    # within ten steps and sooner or later flips an accept (scratch/diag_adaptive_pair.py prints such a pair attempt by
    # attempt).  Two solves at the default rtol = atol = 1e-4 then differ by what the tolerances allow per accepted step,
    # (1e-4 + 1e-4 |lp|) x ~10 steps; measured between f32 / bf16x2 / bf16x3 and the two controllers: 2e-5 .. 2.7e-4.
    assert abs(sm.last_solver_stats["attempts"] - stats["attempts"]) <= 3 and _logp_err(lpd, lp32.cpu()) < 5e-4


@pytest.mark.gpu
def test_split_wide_state_divergence_modes_raise(built_library):
    """bf16x2, 17-32 dimensions: state-only kernels (BASELINE config 5 is a sampler); log-densities of such states under
    precision= raise instead of switching arithmetic (round 3 froze the family at what the configurations reach)."""
    sm, _, _ = _seeded(20, 2, [128, 128], "VPSDE", True, 61, "bf16x2")
    torch.manual_seed(3)
    x0, cond = torch.randn(50, 20) * 0.7, torch.randn(50, 2)
    opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / 12}
    for hutch in (False, True):
        sm.hutch = hutch
        with pytest.raises(NotImplementedError, match="bf16x2"):
            sm.log_prob(x0.to(DEV), conditional=cond.to(DEV), method="rk4", options=opts)
    sm.precision = "f32"
    assert sm.log_prob(x0.to(DEV), conditional=cond.to(DEV), method="rk4", options=opts).shape == (50, 1)


WIDE_STATE = {
    "c5_32d_c8_ve_4x256": (32, 8, [256] * 4, "VESDE", False, "rk4", 25, 300),
    "d20_subvp_ragged": (20, 0, [100, 200], "SUBVPSDE", False, "heun3", 20, 129),
    "d17_c16_vp_one_layer": (17, 16, [256], "VPSDE", True, "euler", 40, 97),
    "d31_c3_four_layers": (31, 3, [128] * 4, "VPSDE", False, "rk4_classic", 10, 65),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(WIDE_STATE))
def test_split_states_of_17_to_32_dimensions(name, built_library):
    """precision="bf16x2" for states of 17-32 dimensions (BASELINE config 5's shape included): two k-steps in the first
    layer (state, then conditional inputs), two row tiles in the output layer, four stage slots on chip.  Sampling and
    Euler-Maruyama (injected stream; in-kernel noise against the numpy Philox restatement and across launch cuts)
    against the oracle at the family's tolerances; methods with more than four stages raise."""
    from tests._philox import normals
    Dm, C, units, sde_name, no_sigma, method, nsteps, B = WIDE_STATE[name]
    sm, so32, so64 = _seeded(Dm, C, units, sde_name, no_sigma, 41, "bf16x2")
    assert "_d2_" in _native.kernel_name(sm._net().plan(0)) and sm._net().stage_slots(0) == 4
    torch.manual_seed(77)
    base = torch.randn(B, Dm) * (float(sm.sde.sigma_max) if hasattr(sm.sde, "sigma_max") else 1.0)
    cond = torch.randn(B, C) if C else None
    cd = None if cond is None else cond.to(DEV)
    opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / nsteps}
    z = base / (float(sm.sde.sigma_max) if hasattr(sm.sde, "sigma_max") else 1.0)
    got, _ = sm.sample_ode_from_base(z.to(DEV), conditional=cd, method=method, options=opts)
    ref64 = so64.sample_ode_from_base(z.double(), None if cond is None else cond.double(), method, opts).float()
    assert _state_err(got, ref64) < STATE_TOL, name
    steps = 30
    draws = [torch.randn(B, Dm) for _ in range(steps)]
    it = iter(draws)
    em = sm._sample_sde_from(base.to(DEV), lambda like: next(it).to(DEV), cd, steps)
    assert _state_err(em, so32.sample_sde(base, draws, cond, steps=steps)) < STATE_TOL, name
    seed, off = 1357, 90
    a = sm._sample_sde_from(base.to(DEV), None, cd, 9, rng=(seed, off))
    zz = torch.from_numpy(normals(seed, off, B, Dm, list(range(9))))
    it = iter(zz)
    b = sm._sample_sde_from(base.to(DEV), lambda like: next(it).to(DEV), cd, 9)
    assert _state_err(a, b.cpu()) < STATE_TOL
    cut = B // 3
    parts = [sm._sample_sde_from(base[lo:hi].to(DEV), None, None if cd is None else cd[lo:hi], 9, rng=(seed, off + lo))
             for lo, hi in ((0, cut), (cut, B))]
    assert torch.equal(torch.cat(parts), a)
    with pytest.raises(NotImplementedError, match="stage"):
        sm.sample_ode_from_base(z.to(DEV), conditional=cd, method="dopri5_fixed", options=opts)
    with pytest.raises(NotImplementedError, match="stage"):
        sm.sample_ode_from_base(z.to(DEV), conditional=cd)                  # adaptive dopri5: seven slots
    # an embedded pair that fits the four slots (bosh3: four with its FSAL stage; on the random sub-VP / VP networks of the
    # other cases bosh3 meets a non-finite error estimate with the f32 kernels as well: fehlberg2 there)
    am = "bosh3" if name.startswith("c5") else "fehlberg2"

    def adaptive(prec):
        sm.precision = prec
        try:
            return sm.sample_ode_from_base(z.to(DEV), conditional=cd, method=am, atol=1e-6, rtol=1e-6)[0], dict(sm.last_solver_stats)
        except RuntimeError as e:              # random stiff networks: a non-finite error estimate stops torchdiffeq the same way
            return str(e), None
    x32, st32 = adaptive("f32")
    xb, stb = adaptive("bf16x2")
    if isinstance(x32, str):
        assert isinstance(xb, str) and ("underflow" in xb or "non-finite" in xb), (x32, xb)
    else:
        assert stb == st32                                               # the same accept / reject sequence
        rb = so64.sample_ode_from_base(z.double(), None if cond is None else cond.double(), am, None, atol=1e-6, rtol=1e-6).float()
        assert _state_err(xb, rb) < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B", [1, 31, 32, 33, 127, 128, 129, 1000])
def test_split_ragged_batches(B, prec, built_library):
    sm, so32, _ = _seeded(16, 0, [64, 64], "VPSDE", True, 15, prec)
    torch.manual_seed(B)
    base = torch.randn(B, 16)
    opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / 8}
    got, _ = sm.sample_ode_from_base(base.to(DEV), method="rk4", options=opts)
    assert _state_err(got, so32.sample_ode_from_base(base, None, "rk4", opts)) < STATE_TOL
    sm.hutch = True
    if prec not in LOGP_PRECS:
        return _logp_raises(lambda: sm.log_prob(base.to(DEV), method="euler", options=opts))
    lp = sm.log_prob(base.to(DEV), method="euler", options=opts)
    assert _logp_err(lp, so32.log_prob(base, None, "euler", opts, "hutch", sm.e.cpu())) < LOGP_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("prec", PRECS)
def test_split_flows_and_wrappers(prec, built_library):
    torch.manual_seed(21)
    f = F.ConditionalODEFlow(target_dimension=16, conditional_dimension=6, hidden_units=[256, 256, 256],
                             target_shift=torch.randn(16), target_scale=torch.rand(16) + 0.5,
                             conditional_shift=torch.randn(6), conditional_scale=torch.rand(6) + 0.5).eval()
    sd = {k: v.detach().clone() for k, v in f.state_dict().items()}
    fo64 = flow_oracle(sd, torch.float64)
    f = f.to(DEV)
    f.precision = prec
    xT, cond = torch.randn(200, 16), torch.randn(200, 6) * 2
    opts = {"step_size": 1.0 / 40}
    got = f.sample(xT.to(DEV), cond.to(DEV), method="rk4", options=opts)               # output affine in the epilogue
    assert _state_err(got, fo64.sample(xT.double(), cond.double(), "rk4", opts).float()) < STATE_TOL
    x = xT[:48] * f.target_scale.cpu() + f.target_shift.cpu()
    torch.manual_seed(3)
    if prec not in LOGP_PRECS:
        _logp_raises(lambda: f.log_prob(x.to(DEV), cond[:48].to(DEV), method="rk4", options=opts, hutchinson=True))
        ga = f.sample(xT.to(DEV), cond.to(DEV))                                        # the default call: adaptive dopri5
        assert _state_err(ga, fo64.sample(xT.double(), cond.double(), "dopri5", None, atol=1e-9, rtol=1e-7).float()) < 2e-4
        return
    lp = f.log_prob(x.to(DEV), cond[:48].to(DEV), method="rk4", options=opts, hutchinson=True)
    f.precision = "f32"
    torch.manual_seed(3)
    lp32 = f.log_prob(x.to(DEV), cond[:48].to(DEV), method="rk4", options=opts, hutchinson=True)
    assert _logp_err(lp, lp32.cpu()) < LOGP_TOL
    # what the family does not do raises instead of switching arithmetic silently
    f.precision = prec
    lpe = f.log_prob(x.to(DEV), cond[:48].to(DEV), method="rk4", options=opts)         # exact trace: 16 unit tangents, two passes
    assert _logp_err(lpe, fo64.log_prob(x.double(), cond[:48].double(), "rk4", opts).float()) < LOGP_TOL
    # the reference's default call -- adaptive dopri5, one launch per attempted step -- on the split kernels
    ga = f.sample(xT.to(DEV), cond.to(DEV))
    assert _state_err(ga, fo64.sample(xT.double(), cond.double(), "dopri5", None, atol=1e-9, rtol=1e-7).float()) < 2e-4
    lpa = f.log_prob(x.to(DEV), cond[:48].to(DEV), hutchinson=True)                     # adaptive, Hutchinson probe
    assert lpa.shape == (48,) and torch.isfinite(lpa).all()


@pytest.mark.gpu
@pytest.mark.parametrize("prec", PRECS)
def test_split_full_size_properties_and_speed(prec, built_library):
    """BASELINE config 2 at 2^20: determinism, batch-shape invariance (bitwise), oracle on a subsample, agreement
    with the f32 kernels.  The reason the family exists is speed: the kernel that served the solve is asserted by NAME and
    the HIP-event times of both arithmetics are printed -- the ratio is bench.py's to report (split_precision_record), a
    wall-clock assertion here would depend on who else is on the card."""
    from flowfusion_amd import _native
    sm, so32, so64 = _seeded(16, 0, [256] * 4, "VPSDE", True, 17, prec)
    B = 1 << 20
    opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / 100}
    g = torch.Generator(device=DEV).manual_seed(1234)
    z = torch.randn(B, 16, device=DEV, generator=g)
    x, _ = sm.sample_ode_from_base(z, method="rk4", options=opts)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    x2, _ = sm.sample_ode_from_base(z, method="rk4", options=opts)
    e1.record()
    torch.cuda.synchronize()
    t_split = e0.elapsed_time(e1) * 1e-3
    assert "split" in _native.kernel_name(sm._net().plan(0))
    assert torch.equal(x, x2) and torch.isfinite(x).all()
    for sl in (slice(0, 200), slice(B // 2 + 5, B // 2 + 77), slice(B - 130, B)):
        xs, _ = sm.sample_ode_from_base(z[sl].contiguous(), method="rk4", options=opts)
        assert torch.equal(xs, x[sl])
    idx = torch.arange(0, B, B // 64)
    ref = so64.sample_ode_from_base(z[idx].cpu().double(), None, "rk4", opts).float()
    assert _state_err(x[idx], ref) < STATE_TOL
    sm.precision = "f32"
    y, _ = sm.sample_ode_from_base(z, method="rk4", options=opts)
    e0.record()
    sm.sample_ode_from_base(z, method="rk4", options=opts)
    e1.record()
    torch.cuda.synchronize()
    t_f32 = e0.elapsed_time(e1) * 1e-3
    assert "split" not in _native.kernel_name(sm._net().plan(0))
    assert ((x - y).abs().max() / y.abs().max()).item() < STATE_TOL
    print(f"\n[split] 2^20 x 100-step RK4: {prec} {t_split * 1e3:.1f} ms ({B / t_split:.3g} samples/s), "
          f"f32 {t_f32 * 1e3:.1f} ms ({B / t_f32:.3g} samples/s), speed-up {t_f32 / t_split:.2f}x (HIP events)")


@pytest.mark.gpu
def test_split_random_shapes_against_oracle(built_library):
    """Seeded sweep over what the split-precision family is compiled for -- both options, on-chip widths 128 and 256, states
    of 1-32 dimensions (bf16x2 beyond 16: state only), 0-16 conditional inputs, 1-4 ragged hidden layers, the three SDEs, fixed-grid
    methods within the kernel's stage slots, batch sizes around the tile sizes -- sampling, Hutchinson / exact-trace
    log-density and Euler-Maruyama against the float64 oracle, so that every (width, state tiles, mode, parts) instance is
    reached by some case."""
    import random
    rnd = random.Random(777)
    kernels = set()
    for case in range(40):
        prec = rnd.choice(PRECS)
        wmax = rnd.choice([24, 64, 100, 128, 129, 200, 256])
        depth = rnd.choice([1, 2, 3, 4])
        units = [rnd.randint(max(8, wmax // 2), wmax) for _ in range(depth)]
        units[rnd.randrange(depth)] = wmax
        Dm = rnd.choice([1, 2, 3, 8, 15, 16] + ([17, 20, 31, 32] if prec == "bf16x2" else []))
        C = rnd.choice([0, 0, 1, 5, 16])
        sde_name = rnd.choice(["VPSDE", "VESDE", "SUBVPSDE"])
        no_sigma = rnd.random() < 0.5
        method, nsteps = rnd.choice([("euler", 12), ("midpoint", 8), ("rk4", 6), ("heun3", 6), ("rk4_classic", 5)] +
                                    ([("dopri5_fixed", 4)] if Dm <= 16 else []))
        B = rnd.choice([1, 7, 31, 33, 64, 130])
        sm, _, so64 = _seeded(Dm, C, units, sde_name, no_sigma, 3000 + case, prec)
        torch.manual_seed(5000 + case)
        z = torch.randn(B, Dm)
        cond = torch.randn(B, C) if C else None
        cd = None if cond is None else cond.to(DEV)
        c64 = None if cond is None else cond.double()
        opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / nsteps}
        tag = (case, prec, Dm, C, units, sde_name, no_sigma, method, B)
        x0, _ = sm.sample_ode_from_base(z.to(DEV), conditional=cd, method=method, options=opts)
        assert _state_err(x0, so64.sample_ode_from_base(z.double(), c64, method, opts).float()) < STATE_TOL, tag
        kernels.add(_native.kernel_name(sm._net().plan(0)))
        mode = rnd.choice(["hutch", "exact"])
        if Dm <= 16 and prec in LOGP_PRECS:
            sm.hutch = mode == "hutch"
            xd = torch.randn(B, Dm) * 0.5
            lp = sm.log_prob(xd.to(DEV), conditional=cd, method=method, options=opts)
            e = sm.e.cpu().double() if sm.hutch else None
            # (two-part operands carry 16 significand bits: twice the fp32 kernels' bar, a fifth of north_star's 1e-4)
            tol = LOGP_TOL * (2 if prec == "bf16x2" else 1)
            assert _logp_err(lp, so64.log_prob(xd.double(), c64, method, opts, mode, e).float()) < tol, tag + (mode,)
            kernels.add(_native.kernel_name(sm._net().plan(1 if sm.hutch else 2)))
            sm.hutch = False
        if case % 4 == 0:
            prior = torch.randn(B, Dm) * (float(sm.sde.sigma_max) if hasattr(sm.sde, "sigma_max") else 1.0)
            draws = [torch.randn(B, Dm) for _ in range(10)]
            it = iter(draws)
            em = sm._sample_sde_from(prior.to(DEV), lambda like: next(it).to(DEV), cd, 10)
            so32 = score_oracle(dict(D=Dm, C=C, E=8, units=units, sde=sde_name, sde_kw={}, no_sigma=no_sigma),
                                {k: v.detach().cpu().clone() for k, v in sm.state_dict().items()})
            assert _state_err(em, so32.sample_sde(prior, draws, cond, steps=10)) < STATE_TOL, tag + ("em",)
    assert all(any(f"_h{w}" in k for k in kernels) for w in (128, 256)) and any("_d2_" in k for k in kernels), sorted(kernels)
    assert len(kernels) >= 16, sorted(kernels)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["bf16x2"])          # (the 128-wide three-part instances went with round 3's freeze)
def test_four_slot_twin_is_bitwise_the_seven_slot_kernel(prec, built_library, monkeypatch):
    """128-wide kernels for states of up to 16 dimensions have a twin with four stage slots on chip (two workgroups per CU): the
    launcher picks it when the caller promises a table of at most four slots (ff_ode_args.stage_slots; the front ends pass the
    method's stage count).  Same arithmetic in the same order: bitwise equal results, for sampling, the Hutchinson and
    exact-trace log-density and Euler-Maruyama; FF_SPLIT_NO_TWIN=1 pins the seven-slot kernel.  (Since round 3 the twins are
    compiled with FF_BUILD_FULL only; on the default library both runs take the seven-slot kernel and the test says that the
    promise changes nothing.)"""
    sm, _, _ = _seeded(9, 3, [128, 100, 128], "VPSDE", True, 71, prec)
    assert _native.kernel_name(sm._net().plan(0)) == ("mlp_ode_split2_h128_n3_t0" if prec == "bf16x2" else "mlp_ode_split_h128_n3_t0")
    torch.manual_seed(6)
    z, cond = torch.randn(1000, 9, device=DEV), torch.randn(1000, 3, device=DEV)
    opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / 10}

    def run():
        out = [sm.sample_ode_from_base(z, conditional=cond, method="rk4", options=opts)[0]]
        sm.hutch = True
        torch.manual_seed(12)
        out.append(sm.log_prob(z[:200], conditional=cond[:200], method="heun3", options=opts))
        sm.hutch = False
        out.append(sm.log_prob(z[:100], conditional=cond[:100], method="rk4", options=opts))          # exact trace
        out.append(sm._sample_sde_from(z, None, cond, 12, rng=(99, 0)))
        return out
    twin = run()
    monkeypatch.setenv("FF_SPLIT_NO_TWIN", "1")
    plain = run()
    for a, b in zip(twin, plain):
        assert torch.equal(a, b)
