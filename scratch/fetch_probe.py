"""FETCH_SIZE probe (round 3): the headline launch with base samples from torch.randn (rounds 1-2) and from ff_normal_fill
(round 3), fresh and re-used buffers -- run under `rocprofv3 --pmc FETCH_SIZE` to see what the read side depends on."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from flowfusion_amd import _native
dev = torch.device("cuda", 0)
sm = bench.build_model(dev)
eps = float(sm.sde.epsilon)
opts = {"step_size": (1.0 - eps) / 100}
B = 1 << 20
g = torch.Generator(device=dev).manual_seed(1234)
z1 = torch.randn(B, 16, device=dev, generator=g)
sm.sample_ode_from_base(z1, method="rk4", options=opts)          # launch 1: torch.randn input, first launch of the process
sm.sample_ode_from_base(z1, method="rk4", options=opts)          # launch 2: same input again
z2 = _native.normal_fill(B, 16, 1234, 0, dev)
sm.sample_ode_from_base(z2, method="rk4", options=opts)          # launch 3: normal_fill input
torch.cuda.synchronize()
