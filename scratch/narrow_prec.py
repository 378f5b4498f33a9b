"""precision= on networks narrower than the split family's 256: is it still a win?  (the family pads to width 256)"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from flowfusion_amd.diffusion import MLP, VESDE, VPSDE, ScoreModel
dev = "cuda"
for D, units in ((2, [128] * 3), (16, [128] * 4), (16, [192] * 4), (16, [64] * 3), (16, [256] * 4)):
    torch.manual_seed(0)
    sm = ScoreModel(MLP(D, 0, 8, units), VPSDE(), no_sigma=True).to(dev).eval()
    z = torch.randn(1 << 20, D, device=dev)
    opts = {"step_size": (1 - 1e-3) / 50}
    row = []
    for prec in ("f32", "bf16x3", "bf16x2"):
        sm.precision = prec
        sm.sample_ode_from_base(z[:4096], method="rk4", options=opts)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sm.sample_ode_from_base(z, method="rk4", options=opts)
        torch.cuda.synchronize(); row.append(time.perf_counter() - t0)
    print(f"dim {D} units {units}: f32 {row[0]*1e3:.0f} ms, bf16x3 {row[1]*1e3:.0f} ms ({row[0]/row[1]:.2f}x), bf16x2 {row[2]*1e3:.0f} ms ({row[0]/row[2]:.2f}x)")
