"""Host profile of one notebook-scale default-argument call on the device controller (where do the ~0.85 ms go?)."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowfusion_amd.diffusion import MLP, VESDE, ScoreModel
dev = torch.device("cuda", 0)
torch.manual_seed(0)
nb = ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).eval().to(dev)
z = torch.randn(1000, 2, device=dev)
for _ in range(20):
    nb.sample_ode_from_base(z)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    nb.sample_ode_from_base(z)
torch.cuda.synchronize()
print("mean call ms", 1e3 * (time.perf_counter() - t0) / 200)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    nb.sample_ode_from_base(z)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
