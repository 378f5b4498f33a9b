"""Kernel time of ONE fixed-grid launch against the number of network evaluations in it (16-d VP 4x256, Hutchinson and
state-only, euler = one evaluation per step): separates the per-launch cost (prologue, tail of the last round of tiles)
from the per-evaluation cost, at the batches the adaptive path's attempts run at."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel
dev = torch.device("cuda", 0)
torch.manual_seed(0)
sm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), hutchinson=True).eval().to(dev)
def ms(fn, reps=5):
    fn(); torch.cuda.synchronize(); out = []
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); out.append(a.elapsed_time(b))
    return min(out)
for B in (1 << 16, 1 << 18, 1 << 20):
    z = torch.randn(B, 16, device=dev)
    for n in (1, 2, 3, 6, 12, 24, 48):
        eps = float(sm.sde.epsilon)
        o = {"step_size": (1.0 - eps) / n}
        s = ms(lambda: sm.sample_ode_from_base(z, method="euler", options=o))
        l = ms(lambda: sm.solve_odes_forward(z, method="euler", options=o))
        print(f"B=2^{B.bit_length()-1} evals {n:3d}: state {s:8.3f} ms ({s/n:7.3f}/eval)   hutchinson {l:8.3f} ms ({l/n:7.3f}/eval)", flush=True)
