"""A/B (round 4): the f32 kernels as built (hipcc's SLP vectoriser packs some of the SiLU tail into v_pk_*_f32
between MFMAs) against the same sources with -fno-slp-vectorize (FF_BUILD_NOSLP=1 python -m flowfusion_amd.build ->
libflowfusion_amd_noslp.so).  HIP-event ms of one launch, min of 3: headline (config 2), config 3 (Hutchinson), and the
128-wide notebook network (state-only and exact trace)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FF_TAIL_SPLIT"] = "0"
from flowfusion_amd import _native, build
from tests.test_gpu_skew import _launch
from tests.test_gpu_parity import _seeded_score_model
DEV = "cuda"
libs = {"as built": _native.lib(), "-fno-slp-vectorize": _native.load_library(build.variant_lib("noslp"))}


def ms_of(fn):
    fn()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


cases = []
sm, _, _ = _seeded_score_model(16, 0, [256] * 4, "VPSDE", True, 17)
B = 1 << 20
x = torch.randn(B, 16, device=DEV)
probe = torch.sign(torch.randn(B, 16, device=DEV))
opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / 100}
span = torch.tensor([1.0, float(sm.sde.epsilon)])
for mode, name in ((0, "headline: 16-d 4x256, 100-step RK4, 2^20"), (1, "config 3: same, Hutchinson column")):
    tab = sm._ode_table(span, "rk4", opts, mode).to(DEV)
    cases.append((name, sm, mode, x, tab, dict(probe=probe) if mode else {}))
nb, _, _ = _seeded_score_model(2, 0, [128] * 3, "VESDE", False, 3)
xb = torch.randn(B, 2, device=DEV)
o2 = {"step_size": (1.0 - float(nb.sde.epsilon)) / 25}
s2 = torch.tensor([1.0, float(nb.sde.epsilon)])
for mode, name in ((0, "notebook net 2-d 3x128, 25-step RK4, 2^20, state only"), (2, "notebook net, exact trace (2 unit tangents)")):
    tab = nb._ode_table(s2, "rk4", o2, mode).to(DEV)
    cases.append((name, nb, mode, xb, tab, {}))

for name, m, mode, xx, tab, kw in cases:
    row, outs = [], []
    for tag, L in libs.items():
        if mode == 2:
            kw = dict(probe=xx)       # (unused by unit tangents; the helper wants a pointer)
        ms = ms_of(lambda: _launch(L, m, mode, xx, tab, tab.shape[0], **kw))
        outs.append(_launch(L, m, mode, xx, tab, tab.shape[0], **kw)[0])
        row.append(f"{tag}: {ms:9.3f} ms")
    print(f"{name:62s} " + "   ".join(row) + f"   bitwise equal: {torch.equal(outs[0], outs[1])}", flush=True)
