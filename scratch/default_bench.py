"""Default-argument calls (adaptive dopri5, exact trace) at the notebook's scale: where does the time go?"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from flowfusion_amd.diffusion import MLP, VESDE, VPSDE, ScoreModel
dev = 'cuda'
torch.manual_seed(0)
for D, units, B in ((2, [128] * 3, 50000), (16, [256] * 4, 1 << 16)):
    sm = ScoreModel(MLP(D, 0, 8, units), VESDE(), no_sigma=False).to(dev).eval()
    z = torch.randn(B, D, device=dev)
    sm.sample_ode_from_base(z); torch.cuda.synchronize()
    t0 = time.time(); x, _ = sm.sample_ode_from_base(z); torch.cuda.synchronize(); dt = time.time() - t0
    st = sm.last_solver_stats
    print(f"D={D} B={B} sample dopri5: {dt*1e3:8.1f} ms  attempts {st['attempts']} accepted {st['accepted']}  -> {dt*1e3/st['attempts']:.2f} ms/attempt", flush=True)
    xs = x[:B // 4].contiguous()
    sm.log_prob(xs); torch.cuda.synchronize()
    t0 = time.time(); lp = sm.log_prob(xs); torch.cuda.synchronize(); dt = time.time() - t0
    st = sm.last_solver_stats
    print(f"D={D} B={B//4} log_prob dopri5 exact: {dt*1e3:8.1f} ms  attempts {st['attempts']} accepted {st['accepted']}  -> {dt*1e3/st['attempts']:.2f} ms/attempt", flush=True)
