"""Per-call wall clock at small batch sizes: host overhead of the fixed-grid, Euler-Maruyama and flow paths."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel
from flowfusion_amd.flow import ODEFlow
dev = 'cuda'
torch.manual_seed(0)
sm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True, hutchinson=True).to(dev).eval()
f = ODEFlow(16, [256] * 3).to(dev).eval()
opts = {"step_size": (1 - 1e-3) / 100}
def wall(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3
for B in (256, 4096, 65536):
    z = torch.randn(B, 16, device=dev)
    print(f"B={B:6d}  rk4-100 sample {wall(lambda: sm.sample_ode_from_base(z, method='rk4', options=opts)):7.2f} ms"
          f"   hutch log_prob {wall(lambda: sm.log_prob(z, method='rk4', options=opts)):7.2f} ms"
          f"   EM-100 {wall(lambda: sm.sample_sde((B, 16), steps=100)):7.2f} ms"
          f"   EM-100 philox {wall(lambda: sm.sample_sde((B, 16), steps=100, noise='philox', seed=1)):7.2f} ms"
          f"   flow rk4-20 {wall(lambda: f.sample(z, method='rk4', options={'step_size': 0.05})):7.2f} ms"
          f"   flow default {wall(lambda: f.sample(z)):7.2f} ms", flush=True)
