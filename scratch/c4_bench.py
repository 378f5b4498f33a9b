import torch, sys, os, time
sys.path.insert(0, '/root/repo')
from flowfusion_amd.flow import ODEFlow
from flowfusion_amd.diffusion import MLP, VESDE, ScoreModel
from flowfusion_amd import _native
torch.manual_seed(0)
dev='cuda'
def timeit(fn, n=2):
    fn(); torch.cuda.synchronize()
    ts=[]
    for _ in range(n):
        s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return min(ts)
# config 4: 64-d flow, 5x512, 200-step fixed dopri (6 evals/step) ; per-GPU share of 2^22 over 8 GPUs = 2^19
f = ODEFlow(64, [512]*5).to(dev).eval()
net = f._net()
for method, nsteps, B in (("rk4", 20, 1<<17), ("dopri5_fixed", 200, 1<<16)):
    tab = f._table(torch.tensor([1.0, 0.0]), method, {"step_size": 1.0/nsteps}, 0).to(dev)
    x = torch.randn(B, 64, device=dev)
    ms = timeit(lambda: net.integrate(x, tab, 0))
    mac = 65*512 + 4*512*512 + 512*64
    flop = 2.0*mac*tab.shape[0]*B
    print(f"C4 {method} {nsteps} steps B={B}: {ms:.1f} ms  {B/ms*1e3:.0f} samples/s  {flop/ms/1e9:.1f} TFLOP/s kernel={_native.lib().ff_kernel_name(net.plan(0).kernel_id).decode()}", flush=True)
# config 5: conditional 32-d VE, 4x256, C=8, 1000-step EM
sm = ScoreModel(MLP(32, 8, 8, [256]*4), VESDE(), no_sigma=False).to(dev).eval()
B = 1<<17
cond = torch.randn(B, 8, device=dev)
t0=time.time(); x = sm.sample_sde((B, 32), conditional=cond, steps=1000); torch.cuda.synchronize(); dt=time.time()-t0
t0=time.time(); x = sm.sample_sde((B, 32), conditional=cond, steps=1000); torch.cuda.synchronize(); dt=time.time()-t0
mac = 48*256 + 3*256*256 + 256*32
print(f"C5 EM 1000 steps B={B}: {dt*1e3:.1f} ms  {B/dt:.0f} samples/s  {2.0*mac*1000*B/dt/1e12:.1f} TFLOP/s (wall, incl. torch noise generation) finite={torch.isfinite(x).all().item()}", flush=True)
for B in (1 << 17, 1 << 20):
    cond = torch.randn(B, 8, device=dev)
    x = sm.sample_sde((B, 32), conditional=cond, steps=1000, noise="philox", seed=1); torch.cuda.synchronize()
    t0=time.time(); x = sm.sample_sde((B, 32), conditional=cond, steps=1000, noise="philox", seed=1); torch.cuda.synchronize(); dt=time.time()-t0
    print(f"C5 EM 1000 steps B={B} in-kernel Philox noise: {dt*1e3:.1f} ms  {B/dt:.0f} samples/s  {2.0*mac*1000*B/dt/1e12:.1f} TFLOP/s (wall) finite={torch.isfinite(x).all().item()}", flush=True)
