import torch, sys
sys.path.insert(0, '/root/repo')
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel
from flowfusion_amd import _native
dev = 'cuda'
B = 1 << 18
opts = {"step_size": (1 - 1e-3) / 100}
for act in (torch.nn.SiLU(), torch.nn.Tanh(), torch.nn.GELU(), torch.nn.Softplus(), torch.nn.ReLU()):
    torch.manual_seed(0)
    sm = ScoreModel(MLP(16, 0, 8, [256]*4, activation=act), VPSDE(), no_sigma=True).eval().to(dev)
    net = sm._net()
    for mode in (0, 1):
        tab = sm._ode_table(torch.tensor([1e-3, 1.0]), "rk4", opts, mode).to(dev)
        x = torch.randn(B, 16, device=dev)
        e = torch.sign(torch.randn(B, 16, device=dev)) if mode == 1 else None
        net.integrate(x, tab, mode, probe=e); torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); net.integrate(x, tab, mode, probe=e); t.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(t))
        ms = min(ts)
        k = _native.kernel_name(net.plan(mode))
        flop = 2 * 206848 * 400 * B * (1, 2)[mode]
        print(f"{type(act).__name__:9s} mode {mode} {k:36s} {ms:8.2f} ms {B/ms*1e3:10.0f} samples/s {flop/ms/1e9:7.2f} TFLOP/s", flush=True)
