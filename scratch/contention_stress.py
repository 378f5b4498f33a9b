"""Determinism under contention: several processes share the card (torch.distributed.run, no collectives needed) and each
repeats (a) the scaled-RMS norm launch, (b) a fixed-grid Hutchinson solve at a cooperative-twin batch, (c) the same at a
one-wavefront batch, (d) a default-argument adaptive solve -- every repeat must return the first repeat's bits."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowfusion_amd import _native
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel
rank = int(os.environ.get("RANK", "0"))
dev = torch.device("cuda", 0)
torch.manual_seed(2)
hm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True, hutchinson=True).eval().to(dev)
REPS = int(os.environ.get("REPS", "300"))

def stress(name, fn, reps=REPS):
    first = fn()
    bad = 0
    worst = 0.0
    for _ in range(reps):
        r = fn()
        if not torch.equal(r, first):
            bad += 1
            worst = max(worst, float((r - first).abs().max() / first.abs().max().clamp_min(1e-30)))
    print(f"rank {rank} {name}: {bad} of {reps} repeats differ (worst relative {worst:.3e})", flush=True)

for n in (3001, 48016 // 16 * 40):
    a = [torch.randn(n * 16, device=dev) for _ in range(3)]
    l = [torch.randn(n, device=dev) for _ in range(3)]
    stress(f"scaled_rms n={n}x16", lambda: torch.tensor(_native.scaled_rms(
        [(a[0], None, a[1], a[2]), (l[0], None, l[1], l[2])], 1e-4, 1e-4, check=a[2])), reps=4 * REPS)
x = torch.randn(3001, 16, device=dev) * 0.8
o = {"step_size": (1.0 - float(hm.sde.epsilon)) / 8}
stress("fixed grid hutchinson B=3001 (cooperative twin)", lambda: hm.log_prob(x, method="rk4", options=o, probe="philox", seed=9))
xl = torch.randn(60000, 16, device=dev) * 0.8
stress("fixed grid hutchinson B=60000 (one-wavefront kernel)", lambda: hm.log_prob(xl, method="rk4", options=o, probe="philox", seed=9), reps=REPS // 3)
stress("adaptive default log_prob B=3001", lambda: hm.log_prob(x, probe="philox", seed=9), reps=REPS // 3)
hm.hutch = False
stress("adaptive default sample B=3001", lambda: hm.sample_ode_from_base(x)[0], reps=REPS // 3)
