"""Non-SiLU activations: TFLOP/s of config 2's network (16-d VP, 4x256) and of a 3x128 one, sampling and Hutchinson mode.
Round 3: A/B of the run-time-choice instantiations (ACT = 9: one per width and mode since round 3; FF_ACT_ANY=1) against the
round-2 compiled-in ones (FF_ACT_ANY=0) -- needs a library built with FF_BUILD_FULL=1 (both sets), e.g.

    FLOWFUSION_AMD_LIB=flowfusion_amd/lib/libflowfusion_amd_full.so python scratch/act_bench.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowfusion_amd import _native                                   # noqa: E402
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel          # noqa: E402

dev = torch.device("cuda", 0)
B = 1 << 20
ACTS = [("SiLU", torch.nn.SiLU()), ("Tanh", torch.nn.Tanh()), ("GELU", torch.nn.GELU()), ("GELU-tanh", torch.nn.GELU(approximate="tanh")),
        ("Softplus", torch.nn.Softplus()), ("ReLU", torch.nn.ReLU()), ("ELU", torch.nn.ELU()), ("Sigmoid", torch.nn.Sigmoid())]
for units in ([256] * 4, [128] * 3):
    x = torch.randn(B, 16, device=dev)
    macs = sum(a * b for a, b in zip([24] + units, units + [16]))
    print(f"--- 16-d VP, hidden {units}")
    for name, act in ACTS:
        row = [f"{name:10s}"]
        for mode in (0, 1):
            for pin in (("1", "0") if name != "SiLU" else (None,)):
                if pin is None:
                    os.environ.pop("FF_ACT_ANY", None)
                else:
                    os.environ["FF_ACT_ANY"] = pin
                torch.manual_seed(0)
                sm = ScoreModel(MLP(16, 0, 8, units, activation=act), VPSDE(), no_sigma=True, hutchinson=(mode == 1)).eval().to(dev)
                try:
                    kname = _native.kernel_name(sm._net().plan(mode))
                except NotImplementedError:
                    row.append(f"mode {mode} any={pin}: no kernel")
                    continue
                opts = {"step_size": (1 - 1e-3) / 25}
                call = (lambda: sm.sample_ode_from_base(x, method="rk4", options=opts)) if mode == 0 else \
                       (lambda: sm.log_prob(x, method="rk4", options=opts))
                call()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                call()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                tf = (2 if mode else 1) * 2.0 * macs * 100 * B / dt / 1e12
                row.append(f"mode {mode} any={pin} {kname.split('mlp_ode_')[1]}: {tf:6.1f} TF")
        print(" | ".join(row), flush=True)
os.environ.pop("FF_ACT_ANY", None)
