import sys, time, torch
sys.path.insert(0, '/root/repo')
from flowfusion_amd.diffusion import MLP, VESDE, ScoreModel
dev = 'cuda'
torch.manual_seed(0)
sm = ScoreModel(MLP(32, 8, 8, [256]*4), VESDE(), no_sigma=False).to(dev).eval()
mac = 48*256 + 3*256*256 + 256*32
for B in (1 << 17, 1 << 19, 1 << 20):
    cond = torch.randn(B, 8, device=dev)
    for noise in ("torch", "philox"):
        kw = dict(noise=noise, seed=1) if noise == "philox" else {}
        sm.sample_sde((B, 32), conditional=cond, steps=200, **kw); torch.cuda.synchronize()
        t0 = time.time(); x = sm.sample_sde((B, 32), conditional=cond, steps=1000, **kw); torch.cuda.synchronize(); dt = time.time() - t0
        print(f"C5 EM 1000 steps B={B} noise={noise:6s}: {dt*1e3:8.1f} ms  {B/dt:9.0f} samples/s  {2.0*mac*1000*B/dt/1e12:6.1f} TFLOP/s (wall)", flush=True)
