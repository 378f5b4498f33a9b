"""Break-even cases of the twin rule, measured directly (FF_TAIL_MAX forces / forbids the tail; FF_COOP pins whole launches)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowfusion_amd.diffusion import MLP, VESDE, VPSDE, ScoreModel
dev = torch.device("cuda", 0)
torch.manual_seed(0)
sm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True).eval().to(dev)
nb = ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).eval().to(dev)
def best(fn, reps=9):
    fn(); torch.cuda.synchronize(); b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, 1e3 * (time.perf_counter() - t0))
    return b
def ab(name, fn, env_a, env_b):
    out = []
    for env in (env_a, env_b, None):
        for k in ("FF_TAIL_MAX", "FF_COOP", "FF_TAIL_SPLIT"):
            os.environ.pop(k, None)
        if env: os.environ.update(env)
        out.append(best(fn))
    print(f"{name}: {env_a} {out[0]:.3f} ms | {env_b} {out[1]:.3f} ms | launcher's choice {out[2]:.3f} ms", flush=True)
x = torch.randn(50000, 2, device=dev) * 0.5
ab("notebook log_prob 50,000 (exact trace; 10,000 tiles, leftover 784 of 3072)", lambda: nb.log_prob(x), {"FF_TAIL_MAX": "0"}, {"FF_TAIL_MAX": "4096"})
for B in (15360 + 3000, 15360 + 3800, 15360 * 2 + 3500):
    xx = torch.randn(B, 2, device=dev) * 0.5
    ab(f"2-d exact log_prob rk4 B={B}", lambda: nb.log_prob(xx, method="rk4", options={"step_size": 0.05}), {"FF_TAIL_MAX": "0"}, {"FF_TAIL_MAX": "4096"})
eps = float(sm.sde.epsilon)
o = {"step_size": (1.0 - eps) / 25}
for B in (20000, 22000, 24576, 26000):
    z = torch.randn(B, 16, device=dev)
    ab(f"16-d sample rk4 B={B} ({B // 16} tiles, one round = 2048)", lambda: sm.sample_ode_from_base(z, method="rk4", options=o), {"FF_COOP": "0"}, {"FF_COOP": "1"})
sm.hutch = True
for B in (9000, 11000, 12288, 13000):
    z = torch.randn(B, 16, device=dev) * 0.5
    ab(f"16-d hutch log_prob rk4 B={B} ({B // 8} tiles)", lambda: sm.log_prob(z, method="rk4", options=o, probe="philox", seed=1), {"FF_COOP": "0"}, {"FF_COOP": "1"})
for B in (2048 * 8 + 5000, 2048 * 8 + 9000, 2048 * 8 + 12500):
    z = torch.randn(B, 16, device=dev) * 0.5
    ab(f"16-d hutch log_prob rk4 B={B} (leftover {(B // 8) % 2048})", lambda: sm.log_prob(z, method="rk4", options=o, probe="philox", seed=1), {"FF_TAIL_MAX": "0"}, {"FF_TAIL_MAX": "4096"})
