import torch, sys
sys.path.insert(0, '/root/repo')
from flowfusion_amd.flow import ODEFlow
from flowfusion_amd import _native
dev = 'cuda'
torch.manual_seed(0)
f = ODEFlow(64, [512]*5).to(dev).eval()
net = f._net()
B = 1 << 17
tab = f._table(torch.tensor([1.0, 0.0]), "rk4", {"step_size": 1.0/20}, 0).to(dev)
x = torch.randn(B, 64, device=dev)
net.integrate(x, tab, 0); torch.cuda.synchronize()
ts = []
for _ in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); net.integrate(x, tab, 0); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
ms = min(ts)
mac = 65*512 + 4*512*512 + 512*64
print(f"C4 rk4 20 steps B={B}: {ms:.1f} ms {2.0*mac*tab.shape[0]*B/ms/1e9:.1f} TFLOP/s", flush=True)
