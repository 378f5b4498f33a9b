import torch, sys, time
sys.path.insert(0, '/root/repo')
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel
dev = 'cuda'
B = 1 << 18
res = []
for nh in (1, 2, 3, 4, 6):
    torch.manual_seed(0)
    sm = ScoreModel(MLP(16, 0, 8, [256] * nh), VPSDE(), no_sigma=True).eval().to(dev)
    net = sm._net()
    opts = {'step_size': (1 - 1e-3) / 50}
    tab = sm._ode_table(torch.tensor([1.0, 1e-3]), 'rk4', opts, 0).to(dev)
    z = torch.randn(B, 16, device=dev)
    net.integrate(z, tab, 0); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); net.integrate(z, tab, 0); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ms = min(ts)
    n_evals = tab.shape[0]
    waves_per_simd = B / 32 / 1024       # sequential waves per SIMD
    us_per_eval = ms * 1e3 / n_evals / waves_per_simd
    res.append((nh, ms, us_per_eval))
    print(f"n_hidden={nh}: {ms:.2f} ms, {us_per_eval:.3f} us per eval per wave", flush=True)
(n0, _, u0), (n1, _, u1) = res[0], res[-1]
per_layer = (u1 - u0) / (n1 - n0)
print(f"per hidden->hidden layer: {per_layer:.3f} us  (ideal 1024 MFMA x 64 cyc @2.38GHz = {1024*64/2380:.3f} us)")
print(f"layer1 + out + bookkeeping (n_hidden=1): {u0:.3f} us (ideal {(64+128)*64/2380:.3f} us)")
