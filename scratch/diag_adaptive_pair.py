"""Diagnostic (round 3): default log_prob (dopri5 + exact trace) of the 2-D VE notebook model under the device and the host
step controller, attempt by attempt: (t, dt, error ratio) -- where do the two sequences part, and how close to 1 was the
ratio there?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_split_precision import _seeded, EXACT, DEV
from flowfusion_amd import adaptive, device_adaptive
Dm, C, units, sde_name, no_sigma, method, nsteps, B = EXACT["d2_ve_notebook"]
sm, so32, so64 = _seeded(Dm, C, units, sde_name, no_sigma, 57, "f32")
torch.manual_seed(8)
x0 = torch.randn(B, Dm) * 0.7
device_adaptive.TRACE = []
sm.log_prob(x0[:40].to(DEV))
dev_trace = device_adaptive.TRACE
device_adaptive.TRACE = None
host_trace = []
orig = adaptive.Dopri5._norms
def spy(self, terms, check=None):
    r = orig(self, terms, check)
    if check is not None:
        host_trace.append(r[0])
    return r
adaptive.Dopri5._norms = spy
orig_attempt = adaptive.Dopri5._attempt
host_td = []
def spy2(self, t0, dt, t1, *a):
    host_td.append((t0, dt))
    return orig_attempt(self, t0, dt, t1, *a)
adaptive.Dopri5._attempt = spy2
os.environ["FF_HOST_CONTROLLER"] = "1"
sm.log_prob(x0[:40].to(DEV))
print("attempt | device: t_after dt_next ratio | host: t_before dt ratio")
for i in range(max(len(dev_trace), len(host_trace))):
    d = dev_trace[i] if i < len(dev_trace) else None
    h = (host_td[i], host_trace[i]) if i < len(host_trace) else None
    print(i, d, h)
