import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from flowfusion_amd import _native
dev = torch.device("cuda", 0)
sm = bench.build_model(dev)
eps = float(sm.sde.epsilon)
opts = {"step_size": (1.0 - eps) / 100}
sm.hutch = True
torch.manual_seed(99)
xq = torch.randn(128, 16) * 0.9
out = {}
for prec in ("f32", "bf16x3", "bf16x2"):
    sm.precision = prec
    torch.manual_seed(99); torch.randn(128, 16)
    lp = sm.log_prob(xq.to(dev), method="rk4", options=opts).cpu()
    out[prec] = lp
    print(prec, _native.kernel_name(sm._net().plan(1)), lp[:3].view(-1).tolist())
print("x3 vs f32", float((out["bf16x3"] - out["f32"]).abs().max()), "x2 vs f32", float((out["bf16x2"] - out["f32"]).abs().max()))
sm.hutch = False
z = torch.randn(256, 16)
xs = {}
for prec in ("f32", "bf16x3", "bf16x2"):
    sm.precision = prec
    xs[prec], _ = sm.sample_ode_from_base(z.to(dev), method="rk4", options=opts)
print("sample x3 vs f32", float((xs["bf16x3"] - xs["f32"]).abs().max()), "x2 vs f32", float((xs["bf16x2"] - xs["f32"]).abs().max()))
