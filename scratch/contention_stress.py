"""Determinism when the card is shared: several processes (torch.distributed.run, no collectives) each repeat a list of
solves; every repeat must return the first repeat's bits.  This is how the start-up race of the cooperative twin was found
(round 3): alone on the card the wavefronts of a workgroup start together and a missing barrier does not show."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowfusion_amd import flow as Fm
from flowfusion_amd.diffusion import MLP, SUBVPSDE, VESDE, VPSDE, ScoreModel
rank = int(os.environ.get("RANK", "0"))
dev = torch.device("cuda", 0)
REPS = int(os.environ.get("REPS", "40"))

def stress(name, fn, reps=REPS):
    try:
        first = fn()
    except Exception as exc:
        print(f"rank {rank} {name}: raised {type(exc).__name__}: {str(exc)[:80]}", flush=True)
        return
    bad, worst = 0, 0.0
    for _ in range(reps):
        r = fn()
        if not torch.equal(r, first):
            bad += 1
            worst = max(worst, float((r - first).abs().max() / first.abs().max().clamp_min(1e-30)))
    print(f"rank {rank} {'FAIL' if bad else 'ok  '} {name}: {bad} of {reps} repeats differ (worst relative {worst:.3e})", flush=True)

torch.manual_seed(2)
sm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True).eval().to(dev)
eps = float(sm.sde.epsilon)
o = {"step_size": (1.0 - eps) / 10}
for B in (700, 3001, 40000):
    x = torch.randn(B, 16, device=dev) * 0.8
    stress(f"16d vp rk4 sample B={B}", lambda: sm.sample_ode_from_base(x, method="rk4", options=o)[0])
    stress(f"16d vp dopri5_fixed sample B={B}", lambda: sm.sample_ode_from_base(x, method="dopri5_fixed", options=o)[0])
    sm.hutch = True
    stress(f"16d vp rk4 hutch log_prob B={B}", lambda: sm.log_prob(x, method="rk4", options=o, probe="philox", seed=3))
    stress(f"16d vp adaptive hutch log_prob B={B}", lambda: sm.log_prob(x, probe="philox", seed=3), reps=REPS // 2)
    sm.hutch = False
    if B <= 3001:
        stress(f"16d vp rk4 exact log_prob B={B}", lambda: sm.log_prob(x, method="rk4", options=o), reps=REPS // 2)
        stress(f"16d vp adaptive exact log_prob B={B}", lambda: sm.log_prob(x), reps=REPS // 4)
    pr = torch.randn(B, 16, device=dev)
    stress(f"16d vp sample_sde philox B={B}", lambda: sm._sample_sde_from(pr.clone(), None, None, 20, rng=(4, 0)))
    for prec in ("bf16x3", "bf16x2"):
        sm.precision = prec
        stress(f"16d vp rk4 sample {prec} B={B}", lambda: sm.sample_ode_from_base(x, method="rk4", options=o)[0])
    sm.precision = "f32"
# the split-precision family under the adaptive driver and with tangents
for prec in ("bf16x2", "bf16x3"):
    sm.precision = prec
    for B in (700, 40000):
        x = torch.randn(B, 16, device=dev) * 0.8
        stress(f"16d vp adaptive sample {prec} B={B}", lambda: sm.sample_ode_from_base(x * 0.5)[0], reps=REPS // 2)
        if prec == "bf16x2":
            sm.hutch = True
            stress(f"16d vp rk4 hutch log_prob {prec} B={B}", lambda: sm.log_prob(x, method="rk4", options=o, probe="philox", seed=3), reps=REPS // 2)
            stress(f"16d vp adaptive hutch log_prob {prec} B={B}", lambda: sm.log_prob(x, probe="philox", seed=3), reps=REPS // 2)
            sm.hutch = False
            if B == 700:
                stress(f"16d vp adaptive exact log_prob {prec} B={B}", lambda: sm.log_prob(x), reps=REPS // 4)
sm.precision = "f32"
torch.manual_seed(0)
nb = ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).eval().to(dev)
nb.precision = "bf16x2"
z2 = torch.randn(20000, 2, device=dev) * 3
stress("2d ve 3x128 adaptive sample bf16x2 B=20000", lambda: nb.sample_ode_from_base(z2)[0], reps=REPS // 2)
stress("2d ve 3x128 adaptive exact log_prob bf16x2 B=20000", lambda: nb.log_prob(z2 * 0.1), reps=REPS // 4)
nb.precision = "f32"
for B in (1000, 50000):
    z = torch.randn(B, 2, device=dev) * 3
    p = torch.randn(B, 2, device=dev) * 0.5
    stress(f"2d ve adaptive sample B={B}", lambda: nb.sample_ode_from_base(z)[0])
    stress(f"2d ve adaptive exact log_prob B={B}", lambda: nb.log_prob(p), reps=REPS // 2)
    for m in ("bosh3", "adaptive_heun"):
        stress(f"2d ve {m} sample B={B}", lambda: nb.sample_ode_from_base(z, method=m)[0], reps=REPS // 2)
torch.manual_seed(1)
cm = ScoreModel(MLP(32, 8, 8, [256] * 4), VESDE()).eval().to(dev)
for B in (900, 30000):
    c = torch.randn(B, 8, device=dev)
    pr = torch.randn(B, 32, device=dev) * 10
    stress(f"32d c8 ve sample_sde philox B={B}", lambda: cm._sample_sde_from(pr.clone(), None, c, 30, rng=(4, 0)))
    z = torch.randn(B, 32, device=dev) * 5
    stress(f"32d c8 ve adaptive sample B={B}", lambda: cm.sample_ode_from_base(z, conditional=c)[0], reps=REPS // 2)
torch.manual_seed(3)
f = Fm.ODEFlow(64, [512] * 5).to(dev).eval()
for B in (300, 3000):
    xT = torch.randn(B, 64, device=dev)
    stress(f"64d flow 5x512 dopri5_fixed B={B}", lambda: f.sample(xT, method="dopri5_fixed", options={"step_size": 0.1}), reps=REPS // 2)
    stress(f"64d flow 5x512 adaptive sample B={B}", lambda: f.sample(xT), reps=REPS // 2)
for units in ([128], [256], [64, 64, 64], [256] * 3, [128] * 5):          # odd numbers of hidden layers, one layer
    torch.manual_seed(5)
    g = ScoreModel(MLP(4, 0, 8, units), VESDE()).eval().to(dev)
    z = torch.randn(800, 4, device=dev) * 3
    tag = "x".join(str(u) for u in units)
    stress(f"4d ve {tag} adaptive sample B=800", lambda: g.sample_ode_from_base(z)[0], reps=REPS // 2)
    stress(f"4d ve {tag} rk4 exact log_prob B=800", lambda: g.log_prob(z * 0.1, method="rk4", options={"step_size": 0.05}), reps=REPS // 2)
# the other tile shapes' twins: 32-column kernels (width 128 beyond 16 dimensions), width 256 up to 64 dimensions
for Dx, units in ((20, [128] * 3), (40, [256] * 3), (24, [128]), (33, [256, 256])):
    torch.manual_seed(9)
    g = ScoreModel(MLP(Dx, 0, 8, units), VESDE()).eval().to(dev)
    z = torch.randn(700, Dx, device=dev) * 3
    tag = f"{Dx}d ve " + "x".join(str(u) for u in units)
    stress(f"{tag} adaptive sample B=700", lambda: g.sample_ode_from_base(z)[0], reps=REPS // 2)
    stress(f"{tag} rk4 sample B=700", lambda: g.sample_ode_from_base(z, method="rk4", options={"step_size": 0.1})[0], reps=REPS // 2)
    g.hutch = True
    stress(f"{tag} adaptive hutch log_prob B=700", lambda: g.log_prob(z * 0.1, probe="philox", seed=2), reps=REPS // 2)
# the estimators' Jacobian output (jac_out / jac_all: written by wavefront 0 of a cooperative workgroup)
for kw in ({"hutchpp": True, "hpp_rank": 2, "hpp_vecs": 2}, {"xtrace": True, "xt_vecs": 3}):
    torch.manual_seed(7)
    em = ScoreModel(MLP(8, 0, 8, [128] * 3), VESDE(), **kw).eval().to(dev)
    xe = torch.randn(600, 8, device=dev) * 0.3
    def est():
        torch.manual_seed(11)                 # the probes are drawn with torch's generator
        return em.log_prob(xe, method="rk4", options={"step_size": 0.1})
    stress(f"8d ve 3x128 {list(kw)[0]} rk4 log_prob B=600", est, reps=REPS // 4)
fw = Fm.ODEFlow(8, [1024, 1024]).to(dev).eval()            # the wide catch-all (cooperative at every batch)
xT = torch.randn(500, 8, device=dev)
stress("8d flow 2x1024 (wide) adaptive sample B=500", lambda: fw.sample(xT), reps=REPS // 2)
stress("8d flow 2x1024 (wide) exact log_prob rk4 B=500", lambda: fw.log_prob(xT, method="rk4", options={"step_size": 0.1}), reps=REPS // 4)
