"""Where the host time of one attempted adaptive step goes at notebook scale (2-D VE 3x128, 50,000 points)."""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, "/root/repo")
from flowfusion_amd.diffusion import MLP, VESDE, ScoreModel
dev = "cuda"
torch.manual_seed(0)
sm = ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).to(dev).eval()
z = torch.randn(50000, 2, device=dev)
for _ in range(3):
    sm.sample_ode_from_base(z)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); sm.sample_ode_from_base(z); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"sample_ode_from_base: min {min(ts)*1e3:.2f} ms  median {sorted(ts)[5]*1e3:.2f} ms  {sm.last_solver_stats}")
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    sm.sample_ode_from_base(z)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
