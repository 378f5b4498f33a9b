"""Round 4: what the estimator launch costs beside the attempt at solver-sized batches: default-argument log_prob (adaptive
dopri5) of BASELINE config 2's network with the exact trace, Hutch++ (r = m = 1) and XTrace (m = 1, 2), 2^16 points; and the
estimator kernel alone on recorded Jacobians of that size (HIP events)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowfusion_amd import _native
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel
dev = "cuda"
torch.manual_seed(0)
sm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True).eval().to(dev)
B = 1 << 16
x = torch.randn(B, 16, device=dev) * 0.8


def timed(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3


for name, kw in (("exact trace", {}), ("hutch++ r=m=1", dict(hutchpp=True)), ("xtrace m=1", dict(xtrace=True)),
                 ("xtrace m=2", dict(xtrace=True, xt_vector=2)), ("hutch++ r=4 m=4", dict(hutchpp=True, hpp_rank=4, hpp_vector=4))):
    sm.hutchpp, sm.xtrace, sm.hpp_rank, sm.hpp_vector, sm.xt_vector = False, False, 1, 1, 1
    for k, v in kw.items():
        setattr(sm, k, v)
    torch.manual_seed(1)
    ms = timed(lambda: sm.log_prob(x))
    print(f"{name:18s} default log_prob 2^16 x 16-d: {ms:8.1f} ms  {sm.last_solver_stats}", flush=True)

jac = torch.randn(6, B, 16, 16, device=dev)
cases = (("hutchpp", (torch.sign(torch.randn(1, B, 16, device=dev)), torch.sign(torch.randn(1, B, 16, device=dev)))),
         ("xtrace", (torch.sign(torch.randn(2, B, 16, device=dev)),)),
         ("hutchpp", (torch.sign(torch.randn(4, B, 16, device=dev)), torch.sign(torch.randn(4, B, 16, device=dev)))))
for pin in ("0", "1"):
    os.environ["FF_TRACE_GENERIC"] = pin
    print("LDS-tile kernel" if pin == "0" else "general kernel (FF_TRACE_GENERIC=1)", flush=True)
    for kind, probes in cases:
        _native.trace_estimate(jac, kind, probes)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            _native.trace_estimate(jac, kind, probes)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        gb = jac.numel() * 4 / 1e9
        print(f"  ff_trace_estimate {kind} probes {tuple(probes[0].shape)}: {ms:.3f} ms for {gb:.2f} GB of Jacobians = "
              f"{gb / ms * 1e3:.0f} GB/s algorithmic (one read), {gb / ms * 1e3 / 8000:.3f} of the 8 TB/s HBM roofline", flush=True)
os.environ.pop("FF_TRACE_GENERIC")
# 32 dimensions (BASELINE config 5's state): tiles of 32 items
B32 = 1 << 14
jac = torch.randn(6, B32, 32, 32, device=dev)
S32, G32 = (torch.sign(torch.randn(1, B32, 32, device=dev)) for _ in range(2))
for pin in ("0", "1"):
    os.environ["FF_TRACE_GENERIC"] = pin
    _native.trace_estimate(jac, "hutchpp", (S32, G32))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        _native.trace_estimate(jac, "hutchpp", (S32, G32))
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    gb = jac.numel() * 4 / 1e9
    print(f"32-d, r = m = 1, 6 x 2^14 items, {'LDS-tile kernel (32-item tiles)' if pin == '0' else 'general kernel'}: {ms:.3f} ms = "
          f"{gb / ms * 1e3:.0f} GB/s algorithmic, {gb / ms * 1e3 / 8000:.3f} of the HBM roofline", flush=True)
os.environ.pop("FF_TRACE_GENERIC")
