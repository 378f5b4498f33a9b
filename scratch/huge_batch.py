"""Index arithmetic beyond 2^31 bytes / 2^31 elements: one huge batch against slices of it solved on their own (bitwise)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowfusion_amd import _native
from flowfusion_amd.diffusion import MLP, VPSDE, VESDE, ScoreModel
dev = torch.device("cuda", 0)
torch.manual_seed(0)
sm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True).eval().to(dev)
eps = float(sm.sde.epsilon)
for logB in (25, 26):
    B = 1 << logB
    z = _native.normal_fill(B, 16, 7, 0, dev)                 # 2^25 x 16 x 4 B = 2 GiB: offsets cross 2^31
    t0 = time.perf_counter()
    x, _ = sm.sample_ode_from_base(z, method="euler", options={"step_size": (1.0 - eps) / 2})
    torch.cuda.synchronize(); print(f"B=2^{logB}: fixed grid {time.perf_counter() - t0:.2f} s", flush=True)
    ok = True
    for lo in (0, B // 2 - 5, B // 2 + 12345, B - 1000):
        part, _ = sm.sample_ode_from_base(z[lo:lo + 1000].contiguous(), method="euler", options={"step_size": (1.0 - eps) / 2})
        ok &= bool(torch.equal(part, x[lo:lo + 1000]))
    print("  slices equal:", ok, " prior rows equal:", bool(torch.equal(_native.normal_fill(100, 16, 7, B - 100, dev), z[B - 100:])), flush=True)
    # streaming helpers over > 2^31 bytes
    vals = _native.scaled_rms([(x, None, z, x)], 1e-3, 1e-3, check=x)
    ref = float(((x.double() / (1e-3 + 1e-3 * torch.max(z.abs(), x.abs()).double())) ** 2).mean().sqrt())
    print("  scaled_rms", vals[0], ref, abs(vals[0] - ref) / ref < 1e-5, flush=True)
    out = torch.empty_like(x)
    _native.stage_combine(out, x, [z], [0.5], 1.0)
    print("  stage_combine tail ok:", bool(torch.equal(out[-1000:], x[-1000:] + 0.5 * z[-1000:])), flush=True)
    del out
    if logB == 25:
        # adaptive (device controller): work buffers of 2 GiB each
        sm2 = ScoreModel(MLP(16, 0, 8, [256] * 4), VESDE(), no_sigma=False).eval().to(dev)
        t0 = time.perf_counter()
        y, _ = sm2.sample_ode_from_base(z * 3)
        torch.cuda.synchronize(); print(f"  adaptive {time.perf_counter() - t0:.2f} s", sm2.last_solver_stats, flush=True)
        lo = B - 2000
        p, _ = sm2.sample_ode_from_base((z[lo:] * 3).contiguous(), options={"first_step": None})
        print("  adaptive tail finite:", bool(torch.isfinite(y[lo:]).all()), " rel diff to tail solved alone:",
              float((p - y[lo:]).abs().max() / y[lo:].abs().max()), flush=True)
        del y, sm2
    del x, z
    torch.cuda.empty_cache()
