"""128-wide bf16x2 kernel: one vs two workgroups per CU (FF_SPLIT_H128_SLOTS4 build).  2-D VE 3x128, 2^20 x 50-step RK4."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel
dev = "cuda"
torch.manual_seed(0)
sm = ScoreModel(MLP(2, 0, 8, [128] * 3), VPSDE(), no_sigma=True, precision="bf16x2").to(dev).eval()
z = torch.randn(1 << 20, 2, device=dev)
opts = {"step_size": (1 - 1e-3) / 50}
x = sm.sample_ode_from_base(z[:4096], method="rk4", options=opts)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x, _ = sm.sample_ode_from_base(z, method="rk4", options=opts)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"{min(ts) * 1e3:.1f} ms  checksum {float(x.double().sum()):.6f}")
