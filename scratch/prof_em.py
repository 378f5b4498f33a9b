import sys, time, torch, cProfile, pstats
sys.path.insert(0, '/root/repo')
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel
dev = 'cuda'
torch.manual_seed(0)
sm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True).to(dev).eval()
for kw in (dict(noise="philox", seed=1), {}):
    sm.sample_sde((4096, 16), steps=100, **kw); torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    sm.sample_sde((4096, 16), steps=100, **kw); torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(10)
