import torch, sys, os
sys.path.insert(0, '/root/repo')
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel
from flowfusion_amd import _native
torch.manual_seed(0)
dev='cuda'
B = 1 << 18
sm = ScoreModel(MLP(16, 0, 8, [256]*4), VPSDE(), no_sigma=True, hutchinson=True).eval().to(dev)
net = sm._net()
opts = {"step_size": (1 - 1e-3) / 100}
for mode, name in ((0, "sample"), (1, "hutch log_prob"), (2, "exact log_prob")):
    tab = sm._ode_table(torch.tensor([1e-3, 1.0]), "rk4", opts, mode).to(dev)
    if mode == 2:
        B = 1 << 16
    x = torch.randn(B, 16, device=dev)
    e = torch.sign(torch.randn(B, 16, device=dev)) if mode == 1 else None
    net.integrate(x, tab, mode, probe=e); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); net.integrate(x, tab, mode, probe=e); t.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(t))
    ms = min(ts)
    k = _native.lib().ff_kernel_name(net.plan(mode).kernel_id).decode()
    flop = 2 * 206848 * 400 * B * (1, 2, 17)[mode]
    print(f"{name:16s} {k:34s} {ms:8.2f} ms  {B/ms*1e3:10.0f} samples/s  {flop/ms/1e9:7.2f} TFLOP/s (tangent FLOPs counted)", flush=True)
