"""Hutch++ / XTrace log_prob on a fixed grid: wall time (one launch per tangent pass with every row's Jacobian recorded)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
for kw in (dict(hutchpp=True, hpp_rank=2, hpp_vecs=2), dict(xtrace=True, xt_vecs=4)):
    sm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True, **kw).eval().to(dev)
    opts = {"step_size": (1 - 1e-3) / 100}
    for B in (1000, 16384):
        x = torch.randn(B, 16, device=dev) * 0.8
        sm.log_prob(x, method="rk4", options=opts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lp = sm.log_prob(x, method="rk4", options=opts)
        torch.cuda.synchronize()
        print(f"{list(kw)[0]:8s} B={B:6d} 100-step RK4: {1e3 * (time.perf_counter() - t0):8.1f} ms  finite={bool(torch.isfinite(lp).all())}")
