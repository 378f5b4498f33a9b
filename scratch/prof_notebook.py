"""Host-side profile of the reference notebook's calls (2-D VE, 3x128, 50,000 points, default arguments)."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from flowfusion_amd.diffusion import MLP, VESDE, ScoreModel  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
sm = ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
z = torch.randn(B, 2, device=dev)
x0 = torch.randn(B, 2, device=dev) * 0.5
for name, fn in (("sample_ode_from_base", lambda: sm.sample_ode_from_base(z)), ("log_prob", lambda: sm.log_prob(x0))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    print(f"== {name}: {1e3 * (time.perf_counter() - t0):.2f} ms, {sm.last_solver_stats}")
    pr = cProfile.Profile()
    pr.enable()
    fn()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
