"""128-wide networks: the 32-column one-wavefront-per-SIMD kernels (FF_TILE=32) against the 16-column two-wavefronts-per-SIMD
instances added in round 3 (default), notebook calls (2-D VE 3x128, 50,000 points, default arguments) and large-batch RK4."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowfusion_amd import _native                                   # noqa: E402
from flowfusion_amd.diffusion import MLP, VESDE, VPSDE, ScoreModel   # noqa: E402

dev = torch.device("cuda", 0)


def best(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        b = min(b, 1e3 * (time.perf_counter() - t0))
    return b


g = torch.Generator(device=dev).manual_seed(4321)
zb = torch.randn(50000, 2, device=dev, generator=g)
xb = torch.randn(50000, 2, device=dev, generator=g) * 0.5
big2 = torch.randn(1 << 20, 2, device=dev, generator=g)
big16 = torch.randn(1 << 20, 16, device=dev, generator=g)
for tile in ("32", "16"):
    os.environ["FF_TILE"] = tile
    torch.manual_seed(0)
    nb = ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).eval().to(dev)
    print(f"FF_TILE={tile}: kernels", _native.kernel_name(nb._net().plan(0)), _native.kernel_name(nb._net().plan(2)))
    print("  notebook sample_ms", round(best(lambda: nb.sample_ode_from_base(zb)), 3), nb.last_solver_stats)
    print("  notebook log_prob_ms", round(best(lambda: nb.log_prob(xb)), 3), nb.last_solver_stats)
    opts = {"step_size": (1.0 - 1e-5) / 50}
    ms = best(lambda: nb.sample_ode_from_base(big2, method="rk4", options=opts), 3)
    macs = 10 * 128 + 2 * 128 * 128 + 128 * 2
    print(f"  2-D 3x128, 2^20 x 50-step RK4: {ms:.1f} ms, {2.0 * macs * 200 * (1 << 20) / ms / 1e9:.1f} TFLOP/s")
    nb.hutch = True
    ms = best(lambda: nb.log_prob(big2, method="rk4", options=opts), 3)
    print(f"  ... Hutchinson log_prob: {ms:.1f} ms, {2 * 2.0 * macs * 200 * (1 << 20) / ms / 1e9:.1f} TFLOP/s")
    torch.manual_seed(0)
    sm = ScoreModel(MLP(16, 0, 8, [128] * 4), VPSDE(), no_sigma=True).eval().to(dev)
    opts = {"step_size": (1.0 - 1e-3) / 50}
    ms = best(lambda: sm.sample_ode_from_base(big16, method="rk4", options=opts), 3)
    macs = 24 * 128 + 3 * 128 * 128 + 128 * 16
    print(f"  16-d 4x128 ({_native.kernel_name(sm._net().plan(0))}), 2^20 x 50-step RK4: {ms:.1f} ms, {2.0 * macs * 200 * (1 << 20) / ms / 1e9:.1f} TFLOP/s")
os.environ.pop("FF_TILE", None)
