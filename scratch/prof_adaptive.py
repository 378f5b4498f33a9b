import sys, time, torch, cProfile, pstats
sys.path.insert(0, '/root/repo')
from flowfusion_amd.diffusion import MLP, VESDE, ScoreModel
dev = 'cuda'
torch.manual_seed(0)
sm = ScoreModel(MLP(16, 0, 8, [256] * 4), VESDE(), no_sigma=False).to(dev).eval()
z = torch.randn(65536, 16, device=dev)
sm.sample_ode_from_base(z); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
x, _ = sm.sample_ode_from_base(z); torch.cuda.synchronize()
pr.disable()
print(sm.last_solver_stats)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
