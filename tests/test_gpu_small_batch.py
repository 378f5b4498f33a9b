"""GPU tier: the cooperative (small-batch) twins of the f32 kernels.  Below about half a chip's worth of tiles the
launcher hands a batch to a kernel that gives each tile to a whole WORKGROUP (the rows of every layer split over its
four wavefronts, activations exchanged through LDS) instead of one wavefront.  It uses the same packed weights and
performs the same fp32 FMA chains in the same order, so the results must equal the one-wavefront kernel's BIT FOR BIT
-- in every mode (state, Hutchinson, exact trace, Euler-Maruyama, adaptive steps, Jacobian output), for every
kernel layout that has a twin.  FF_COOP=0 / 1 pins the choice per launch."""
import pytest
import torch

from tests.test_gpu_parity import DEV, _seeded_score_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(built_library):
    assert torch.cuda.is_available(), "the gpu tier needs a GPU"


def _both(monkeypatch, fn):
    out = []
    for pin in ("0", "1"):
        monkeypatch.setenv("FF_COOP", pin)
        out.append(fn())
    monkeypatch.delenv("FF_COOP")
    return out


SHAPES = [
    (16, 0, [256] * 4, "VPSDE", True),          # 16x16x4, two waves per SIMD (BASELINE config 2)
    (2, 0, [128] * 3, "VESDE", False),          # the demo notebook's network: 32x32x2, width 128
    (32, 8, [256] * 4, "VESDE", False),         # BASELINE config 5 shape
    (5, 3, [128, 100], "SUBVPSDE", False),      # ragged widths, conditional
    (40, 0, [512, 300], "VPSDE", True),         # 512-wide kernels
    (16, 0, [256], "VPSDE", True),              # a single hidden layer
]


@pytest.mark.parametrize("D,C,units,sde,no_sigma", SHAPES)
def test_cooperative_twin_is_bitwise_the_one_wavefront_kernel(D, C, units, sde, no_sigma, monkeypatch):
    sm, so32, _ = _seeded_score_model(D, C, units, sde, no_sigma, 7)
    B = 77
    torch.manual_seed(1)
    base = torch.randn(B, D, device=DEV)
    cond = torch.randn(B, C, device=DEV) if C else None
    opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / 12}
    a, b = _both(monkeypatch, lambda: sm.sample_ode_from_base(base, conditional=cond, method="rk4", options=opts)[0])
    assert torch.equal(a, b)
    # and the oracle agrees (the twin is a kernel of its own, not just a consistent one)
    ref = so32.sample_ode_from_base(base.cpu(), None if cond is None else cond.cpu(), "rk4", opts)
    assert ((b.cpu() - ref).abs().max() / ref.abs().max()).item() < 2e-5
    # Hutchinson: value / tangent column pairs
    sm.hutch = True
    net = sm._net()
    e = torch.sign(torch.randn(B, D, device=DEV))
    tab = sm._ode_table(torch.tensor([float(sm.sde.epsilon), 1.0]), "dopri5_fixed", opts, 1)
    r0, r1 = _both(monkeypatch, lambda: net.integrate(base * 0.5, tab, 1, cond=cond, probe=e))
    assert torch.equal(r0[0], r1[0]) and torch.equal(r0[1], r1[1])
    # exact trace (unit tangents, several passes when D + 1 exceeds the tile)
    sm.hutch = False
    l0, l1 = _both(monkeypatch, lambda: sm.log_prob(base[:20] * 0.5, conditional=None if cond is None else cond[:20],
                                                    method="euler", options=opts))
    assert torch.equal(l0, l1)
    # Euler-Maruyama with a noise buffer and with the in-kernel stream
    noise = torch.randn(9, B, D, device=DEV)
    def em():
        it = iter(noise)
        return sm._sample_sde_from(base, lambda like: next(it), cond, steps=9)
    s0, s1 = _both(monkeypatch, em)
    assert torch.equal(s0, s1)
    p0, p1 = _both(monkeypatch, lambda: sm._sample_sde_from(base, None, cond, 9, rng=(5, 100)))
    assert torch.equal(p0, p1)


def test_cooperative_twin_adaptive_and_jacobian(monkeypatch):
    sm, _, _ = _seeded_score_model(4, 0, [128, 128], "VESDE", False, 51)
    torch.manual_seed(8)
    base = torch.randn(300, 4, device=DEV)
    a, b = _both(monkeypatch, lambda: sm.sample_ode_from_base(base)[0])                   # dopri5: one launch per attempt
    assert torch.equal(a, b)
    x0 = torch.randn(64, 4, device=DEV) * 0.5
    l0, l1 = _both(monkeypatch, lambda: sm.log_prob(x0))                                   # adaptive + exact trace
    assert torch.equal(l0, l1)
    sm.hutchpp, sm.hpp_rank = True, 4                                                       # Jacobian output path
    torch.manual_seed(3)
    j0 = None
    outs = []
    for pin in ("0", "1"):
        monkeypatch.setenv("FF_COOP", pin)
        torch.manual_seed(3)
        outs.append(sm.log_prob(x0, method="rk4", options={"step_size": 0.25}))
    monkeypatch.delenv("FF_COOP")
    assert torch.equal(outs[0], outs[1])


def test_small_batches_take_the_twin_by_default():
    """Default dispatch: at 2048 samples (128 tiles of 16) the launcher takes the cooperative twin -- asked of the
    launcher's own rule (ff_mlp_launch_kind), not inferred from a clock: a wall-clock ratio depends on who else is on the
    card.  The HIP-event times of both kernels are printed for the record; the ratio itself is bench.py's to report
    ("small batch" entry: speedup_of_cooperative_twin).  Bitwise the same results either way."""
    import os
    from flowfusion_amd import _native
    sm, _, _ = _seeded_score_model(16, 0, [256] * 4, "VPSDE", True, 17)
    opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / 100}
    z = torch.randn(2048, 16, device=DEV)
    plan = sm._net().plan(0)
    assert "FF_COOP" not in os.environ
    assert _native.launch_kind(plan, 2048, 0) == _native.LAUNCH_TWIN
    assert _native.launch_kind(plan, 1 << 20, 0) == _native.LAUNCH_ONE_WAVE
    assert _native.launch_kind(plan, (1 << 15) + 300, 0) == _native.LAUNCH_ONE_WAVE_AND_TWIN      # two rounds and a few tiles

    def timed():
        sm.sample_ode_from_base(z, method="rk4", options=opts)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        x, _ = sm.sample_ode_from_base(z, method="rk4", options=opts)
        t1.record()
        torch.cuda.synchronize()
        return x, t0.elapsed_time(t1)

    x_def, ms_def = timed()
    os.environ["FF_COOP"] = "0"
    try:
        assert _native.launch_kind(plan, 2048, 0) == _native.LAUNCH_ONE_WAVE
        x_one, ms_one = timed()
    finally:
        del os.environ["FF_COOP"]
    assert torch.equal(x_def, x_one)
    print(f"\n[small batch] 2048 x 100-step RK4 (HIP events): default = twin {ms_def:.2f} ms, one-wavefront kernel {ms_one:.2f} ms "
          f"({ms_one / ms_def:.2f}x)")


@pytest.mark.parametrize("D,C,units,sde", [(16, 0, [1024, 1024], "VPSDE"), (100, 40, [300, 700], "VESDE"), (64, 50, [512], "VPSDE")])
def test_wide_catch_all_kernels_against_oracle(D, C, units, sde):
    """Beyond 512 wide / 64 dimensions / 32 conditional inputs the plan lands on the wide catch-alls (a tile per
    workgroup at every batch size, hidden operands streamed from LDS): sampling, Hutchinson and exact log-density, and
    Euler-Maruyama against the oracle, and batch-shape invariance across the workgroup boundary."""
    from flowfusion_amd import _native
    from tests.test_gpu_parity import LOGP_TOL, STATE_TOL, _logp_err, _state_err
    sm, so32, so64 = _seeded_score_model(D, C, units, sde, sde == "VPSDE", 29)
    assert _native.kernel_name(sm._net().plan(0)).endswith("_wide")
    B = 70
    torch.manual_seed(3)
    base = torch.randn(B, D)
    cond = torch.randn(B, C) if C else None
    cd = None if cond is None else cond.to(DEV)
    opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / 6}
    x, _ = sm.sample_ode_from_base(base.to(DEV), conditional=cd, method="rk4", options=opts)
    ref = so64.sample_ode_from_base(base.double(), None if cond is None else cond.double(), "rk4", opts).float()
    assert _state_err(x, ref) < STATE_TOL
    xs, _ = sm.sample_ode_from_base(base[10:33].to(DEV), conditional=None if cd is None else cd[10:33].contiguous(),
                                    method="rk4", options=opts)
    assert torch.equal(xs, x[10:33])
    sm.hutch = True
    x0 = base[:24] * 0.5
    c24 = None if cd is None else cd[:24].contiguous()
    torch.manual_seed(4)
    lp = sm.log_prob(x0.to(DEV), conditional=c24, method="midpoint", options=opts)
    ref = so64.log_prob(x0.double(), None if cond is None else cond[:24].double(), "midpoint", opts, "hutch", sm.e.cpu().double()).float()
    assert _logp_err(lp, ref) < LOGP_TOL
    if D <= 64:                                      # exact trace: D unit tangents in passes of 15
        sm.hutch = False
        lp = sm.log_prob(x0[:6].to(DEV), conditional=None if c24 is None else c24[:6].contiguous(), method="euler", options=opts)
        ref = so64.log_prob(x0[:6].double(), None if cond is None else cond[:6].double(), "euler", opts, "exact").float()
        assert _logp_err(lp, ref) < LOGP_TOL
    noise = torch.randn(5, B, D)
    it = iter(noise)
    s = sm._sample_sde_from(base.to(DEV), lambda like: next(it).to(DEV), cd, steps=5)
    assert _state_err(s, so32.sample_sde(base, list(noise), cond, steps=5)) < STATE_TOL


def test_tail_of_a_launch_goes_to_the_twin(monkeypatch):
    """A launch of a few rounds whose leftover tiles are few: the leftover rows run on the cooperative twin as a second
    launch (ff_mlp_ode_launch, "the tail of a launch").  Every mode, with FF_TAIL_SPLIT=0 as the unsplit yardstick: bitwise
    the same outputs -- state, divergence, in-kernel and supplied noise, the adaptive attempt's auxiliary outputs (through
    a default-argument solve), conditional inputs."""
    from flowfusion_amd.diffusion import MLP, VESDE, VPSDE, ScoreModel
    torch.manual_seed(0)
    sm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True).eval().to(DEV)       # 2 wavefronts per SIMD: 2048 tiles at once
    nb = ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).eval().to(DEV)                       # 3 per SIMD: 3072
    cm = ScoreModel(MLP(8, 3, 8, [256] * 2), VESDE()).eval().to(DEV)
    eps = float(sm.sde.epsilon)
    o = {"step_size": (1.0 - eps) / 3}

    def both(fn):
        monkeypatch.setenv("FF_TAIL_SPLIT", "0")
        a = fn()
        monkeypatch.delenv("FF_TAIL_SPLIT")
        return a, fn()

    z = torch.randn(2048 * 16 + 500, 16, device=DEV)                                        # state: 16 samples per tile
    a, b = both(lambda: sm.sample_ode_from_base(z, method="rk4", options=o)[0])
    assert torch.equal(a, b)
    sm.hutch = True
    x = torch.randn(2048 * 8 * 2 + 77, 16, device=DEV) * 0.5                                # Hutchinson: 8 per tile, two rounds + 10 tiles
    a, b = both(lambda: sm.log_prob(x, method="rk4", options=o, probe="philox", seed=3))
    assert torch.equal(a, b)
    sm.hutch = False
    p = torch.randn(3072 * 5 + 301, 2, device=DEV) * 0.3                                    # exact trace, 2-d: 5 samples per tile
    a, b = both(lambda: nb.log_prob(p, method="rk4", options={"step_size": 0.25}))
    assert torch.equal(a, b)
    pr = torch.randn(2048 * 16 + 999, 16, device=DEV)
    a, b = both(lambda: sm._sample_sde_from(pr.clone(), None, None, 12, rng=(4, 123456)))   # in-kernel noise keyed by the global row
    assert torch.equal(a, b)

    def em_torch():
        torch.manual_seed(5)
        return sm.sample_sde((2048 * 16 + 40, 16), steps=6)
    a, b = both(em_torch)                                                                   # supplied noise slabs
    assert torch.equal(a, b)
    c = torch.randn(2048 * 16 + 333, 3, device=DEV)
    zc = torch.randn(2048 * 16 + 333, 8, device=DEV)
    a, b = both(lambda: cm.sample_ode_from_base(zc, conditional=c, method="midpoint", options={"step_size": 0.2})[0])
    assert torch.equal(a, b)
    zn = torch.randn(50000, 2, device=DEV) * 3                                              # the notebook call: 3125 tiles, 53 left over
    a, b = both(lambda: nb.sample_ode_from_base(zn)[0])                                     # adaptive: k1_in and four auxiliary outputs
    assert torch.equal(a, b) and nb.last_solver_stats["accepted"] >= 5
    pn = torch.randn(3072 * 5 + 120, 2, device=DEV) * 0.4
    a, b = both(lambda: nb.log_prob(pn))                                                    # adaptive with the divergence
    assert torch.equal(a, b)
