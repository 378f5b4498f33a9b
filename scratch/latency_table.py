"""Latency of one solve vs batch size, cooperative twin (default dispatch) vs the one-wavefront kernel (FF_COOP=0):
BASELINE config 2 (100-step RK4, 16-dim 4x256) and the reference notebook's own shape (2-D VE, 3x128; default
arguments = adaptive dopri5 + exact trace, demo_diffusion.ipynb:388,467).  Run on the GPU box; prints a table."""
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from flowfusion_amd.diffusion import MLP, VESDE, VPSDE, ScoreModel  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def both(fn):
    out = []
    for pin in (None, "0"):
        if pin is None:
            os.environ.pop("FF_COOP", None)
        else:
            os.environ["FF_COOP"] = pin
        out.append(timed(fn))
    os.environ.pop("FF_COOP", None)
    return out


torch.manual_seed(0)
sm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True).eval().to(dev)
opts = {"step_size": (1.0 - 1e-3) / 100}
print("config 2 (16-dim VP, 4x256, 100-step RK4): batch, default ms, one-wavefront ms, ratio, samples/s (default)")
for B in (256, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
    z = torch.randn(B, 16, device=dev)
    a, b = both(lambda: sm.sample_ode_from_base(z, method="rk4", options=opts))
    print(f"  {B:6d}  {a * 1e3:8.2f}  {b * 1e3:8.2f}  {b / a:5.2f}x  {B / a:10.4g}")
torch.manual_seed(0)
demo = ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).eval().to(dev)
print("notebook shape (2-D VE, 3x128), reference default arguments: batch, call, default ms, one-wavefront ms, ratio")
for B in (1000, 5000, 50000):
    z = torch.randn(B, 2, device=dev)
    x0 = torch.randn(B, 2, device=dev) * 0.5
    for name, fn in (("sample_ode_from_base (dopri5)", lambda: demo.sample_ode_from_base(z)),
                     ("log_prob (dopri5, exact trace)", lambda: demo.log_prob(x0)),
                     ("sample_sde (100 steps)", lambda: demo.sample_sde((B, 2)))):
        a, b = both(fn)
        print(f"  {B:6d}  {name:32s} {a * 1e3:8.2f}  {b * 1e3:8.2f}  {b / a:5.2f}x")
