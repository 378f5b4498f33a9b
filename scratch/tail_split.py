"""The tail of a launch on the cooperative twin (ff_mlp_ode_launch): wall time of one fixed-grid solve with FF_TAIL_SPLIT=0 / 1
at batches a little above a whole number of rounds (16-d VP 4x256, 20-step RK4: 2048 tiles of 16 samples per round)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowfusion_amd.diffusion import MLP, VESDE, VPSDE, ScoreModel
dev = torch.device("cuda", 0)
torch.manual_seed(0)
sm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True).eval().to(dev)
nb = ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).eval().to(dev)
eps = float(sm.sde.epsilon)
o = {"step_size": (1.0 - eps) / 20}
def best(fn, reps=7):
    fn(); torch.cuda.synchronize(); b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, 1e3 * (time.perf_counter() - t0))
    return b
R = 2048 * 16
for B in (R, R + 16, R + 1000, R + 8000, R + 12000, 2 * R + 3000, 3 * R + 500, 8 * R + 2000, 50000):
    z = torch.randn(B, 16, device=dev)
    row = []
    for s in ("0", "1"):
        os.environ["FF_TAIL_SPLIT"] = s
        row.append(best(lambda: sm.sample_ode_from_base(z, method="rk4", options=o)))
    print(f"16-d 4x256 rk4 x20  B={B:7d} ({B / R:5.2f} rounds): unsplit {row[0]:7.2f} ms  split {row[1]:7.2f} ms  ({100 * (row[0] / row[1] - 1):+5.1f} %)", flush=True)
zn = torch.randn(50000, 2, device=dev) * 3
for s in ("0", "1"):
    os.environ["FF_TAIL_SPLIT"] = s
    print(f"notebook sample_ode_from_base, 50,000 x 2-d (3125 tiles, 3072 at once) FF_TAIL_SPLIT={s}: {best(lambda: nb.sample_ode_from_base(zn)):.2f} ms", flush=True)

# how large a leftover is still better off on the twin?  (FF_TAIL_MAX overrides the launcher's bound)
os.environ["FF_TAIL_SPLIT"] = "1"
for rem in (300, 600, 900, 1077, 1300, 1600, 1900, 2040):
    B = R + rem * 16
    z = torch.randn(B, 16, device=dev)
    os.environ["FF_TAIL_MAX"] = "0"
    a = best(lambda: sm.sample_ode_from_base(z, method="rk4", options=o))
    os.environ["FF_TAIL_MAX"] = "4096"
    b = best(lambda: sm.sample_ode_from_base(z, method="rk4", options=o))
    print(f"leftover {rem:5d} tiles after one round: one-wavefront tail {a:6.2f} ms, twin tail {b:6.2f} ms", flush=True)
os.environ.pop("FF_TAIL_MAX")
torch.manual_seed(1)
f5 = __import__("flowfusion_amd.flow", fromlist=["ODEFlow"]).ODEFlow(64, [512] * 5).to(dev).eval()      # width 512: one wavefront per SIMD
R5 = 1024 * 16
for rem in (100, 256, 400, 700, 1000):
    xT = torch.randn(R5 + rem * 16, 64, device=dev)
    os.environ["FF_TAIL_MAX"] = "0"
    a = best(lambda: f5.sample(xT, method="rk4", options={"step_size": 0.25}), 3)
    os.environ["FF_TAIL_MAX"] = "4096"
    b = best(lambda: f5.sample(xT, method="rk4", options={"step_size": 0.25}), 3)
    print(f"5x512 flow, leftover {rem:5d} tiles after one round of 1024: one-wavefront tail {a:6.2f} ms, twin tail {b:6.2f} ms", flush=True)
R3 = 3072 * 16                                                    # width 128: three wavefronts per SIMD
o3 = {"step_size": 0.02}
for rem in (0, 300, 600, 900, 1100, 1500, 1900, 2300, 2700, 3000):
    zz = torch.randn(R3 + rem * 16, 2, device=dev)
    os.environ["FF_TAIL_MAX"] = "0"
    a = best(lambda: nb.sample_ode_from_base(zz, method="rk4", options=o3))
    os.environ["FF_TAIL_MAX"] = "8192"
    b = best(lambda: nb.sample_ode_from_base(zz, method="rk4", options=o3))
    print(f"3x128 (3 wavefronts per SIMD), leftover {rem:5d} tiles after one round of 3072: one-wavefront tail {a:6.2f} ms, twin tail {b:6.2f} ms", flush=True)
