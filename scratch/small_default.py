"""Default-argument calls (adaptive dopri5) at small batches of the notebook model (2-D VE 3x128): device vs host controller."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowfusion_amd.diffusion import MLP, VESDE, ScoreModel
dev = torch.device("cuda", 0)
torch.manual_seed(0)
nb = ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).eval().to(dev)
def best(fn, reps=7):
    fn(); torch.cuda.synchronize(); b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, 1e3 * (time.perf_counter() - t0))
    return b
for B in (100, 1000, 5000, 20000, 50000):
    z = torch.randn(B, 2, device=dev); x = torch.randn(B, 2, device=dev) * 0.5
    row = [f"B={B:6d}"]
    for env in (None, "1"):
        if env: os.environ["FF_HOST_CONTROLLER"] = env
        else: os.environ.pop("FF_HOST_CONTROLLER", None)
        s = best(lambda: nb.sample_ode_from_base(z)); a1 = nb.last_solver_stats["attempts"]
        l = best(lambda: nb.log_prob(x)); a2 = nb.last_solver_stats["attempts"]
        row.append(f"{'host  ' if env else 'device'}: sample {s:6.2f} ms ({a1} att)  log_prob {l:6.2f} ms ({a2} att)")
    print(" | ".join(row), flush=True)
os.environ.pop("FF_HOST_CONTROLLER", None)
