"""Profiling driver for the split-precision kernel: BASELINE config 2 (or 3 with --hutch) at 2^20, a few launches,
nothing else on the GPU.  Run under rocprofv3 (kernel trace / PMC passes) or alone (prints HIP-event times)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel  # noqa: E402

prec = "f32" if "--f32" in sys.argv else ("bf16x2" if "--x2" in sys.argv else "bf16x3")
hutch = "--hutch" in sys.argv
n = 3
dev = torch.device("cuda", 0)
torch.manual_seed(0)
sm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True, precision=prec).eval().to(dev)
eps = float(sm.sde.epsilon)
opts = {"step_size": (1.0 - eps) / 100}
B = 1 << 20
g = torch.Generator(device=dev).manual_seed(1234)
z = torch.randn(B, 16, device=dev, generator=g)
net = sm._net()
mode = 1 if hutch else 0
tab = sm._ode_table(torch.tensor([eps, 1.0] if hutch else [1.0, eps]), "rk4", opts, mode).to(dev)
probe = torch.sign(torch.randn(B, 16, device=dev, generator=g)) if hutch else None
net.integrate(z[:4096].contiguous(), tab, mode, probe=None if probe is None else probe[:4096].contiguous())
torch.cuda.synchronize()
ms = []
for _ in range(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    net.integrate(z, tab, mode, probe=probe)
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
print(f"{prec} {'hutch' if hutch else 'state'}: kernel ms {['%.1f' % m for m in ms]}  -> {B / (min(ms) * 1e-3):.4g} units/s")
