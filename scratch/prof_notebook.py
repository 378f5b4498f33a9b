"""The reference's notebook calls (2-D VE, 3x128, 50,000 points, default arguments) a few times each -- run under
rocprofv3 --kernel-trace --stats to see where an attempted step's time goes (fused attempt / controller / commit)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowfusion_amd.diffusion import MLP, VESDE, ScoreModel          # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
nb = ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).eval().to(dev)
g = torch.Generator(device=dev).manual_seed(4321)
zb = torch.randn(50000, 2, device=dev, generator=g)
xb = torch.randn(50000, 2, device=dev, generator=g) * 0.5
for what, fn in (("sample", lambda: nb.sample_ode_from_base(zb)), ("log_prob", lambda: nb.log_prob(xb))):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(what, "ms:", " ".join(f"{t:.2f}" for t in ts), dict(nb.last_solver_stats), flush=True)
