"""Mid-size batches (the reference's notebook scale): one-wavefront kernel (FF_COOP=0) vs cooperative twin (FF_COOP=1) vs the
launcher's own choice, 2-D VE 3x128 and 16-d VP 4x256, sampling (25-step RK4) and exact-trace log-density, over batch sizes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowfusion_amd.diffusion import MLP, VESDE, VPSDE, ScoreModel          # noqa: E402

dev = torch.device("cuda", 0)


def ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0))
    return best


for label, sm, D in (("2-D VE 3x128", ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).eval().to(dev), 2),
                     ("16-d VP 4x256", ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True).eval().to(dev), 16)):
    opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / 25}
    for what in ("sample", "exact log_prob", "hutch log_prob"):
        print(f"--- {label}, {what}: batch | one-wave ms | coop ms | default ms")
        for B in (4096, 8192, 12288, 16384, 24576, 32768, 50000, 65536, 100000, 131072):
            if what != "sample" and D == 16 and B > 65536:
                continue
            x = torch.randn(B, D, device=dev)
            sm.hutch = what.startswith("hutch")
            fn = (lambda: sm.sample_ode_from_base(x, method="rk4", options=opts)) if what == "sample" else \
                 (lambda: sm.log_prob(x, method="rk4", options=opts))
            row = []
            for pin in ("0", "1", None):
                if pin is None:
                    os.environ.pop("FF_COOP", None)
                else:
                    os.environ["FF_COOP"] = pin
                row.append(ms(fn))
            print(f"{B:7d} | {row[0]:8.3f} | {row[1]:8.3f} | {row[2]:8.3f}", flush=True)
os.environ.pop("FF_COOP", None)
