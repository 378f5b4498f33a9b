"""A/B (round 4): the slope hand-over of the tangent columns by ds_bpermute (general: any number of tangent columns per
sample) against DPP row_shr:1 (Hutchinson pairs only: the value column is the lane to the left) -- FF_BUILD_EXP_DPP=1 builds
libflowfusion_amd_dpp.so.  HIP-event ms of one launch, min of 3."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FF_TAIL_SPLIT"] = "0"
from flowfusion_amd import _native, build
from tests.test_gpu_skew import _launch
from tests.test_gpu_parity import _seeded_score_model
DEV = "cuda"
libs = {"ds_bpermute": _native.lib(), "dpp": _native.load_library(build.variant_lib("dpp"))}


def ms_of(fn):
    fn()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


B = 1 << 20
for name, (D, units, sde, steps) in {"config 3: 16-d 4x256 Hutchinson, 100-step RK4": (16, [256] * 4, "VPSDE", 100),
                                     "notebook net 2-d 3x128 Hutchinson, 25-step RK4": (2, [128] * 3, "VESDE", 25)}.items():
    sm, _, _ = _seeded_score_model(D, 0, units, sde, sde == "VPSDE", 17)
    x = torch.randn(B, D, device=DEV)
    probe = torch.sign(torch.randn(B, D, device=DEV))
    tab = sm._ode_table(torch.tensor([1.0, float(sm.sde.epsilon)]), "rk4", {"step_size": (1.0 - float(sm.sde.epsilon)) / steps}, 1).to(DEV)
    row, outs = [], []
    for tag, L in libs.items():
        ms = ms_of(lambda: _launch(L, sm, 1, x, tab, tab.shape[0], probe=probe))
        outs.append(_launch(L, sm, 1, x, tab, tab.shape[0], probe=probe))
        row.append(f"{tag}: {ms:9.3f} ms")
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2])
    print(f"{name:52s} " + "   ".join(row) + f"   bitwise equal: {same}", flush=True)
