"""Diagnostic (round 4): default log_prob with XTrace / Hutch++ under the device route and the host route, attempt by
attempt: (t, dt, error ratio) -- where do the two sequences part, and how close to 1 was the ratio there?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests._util import load_golden, score_model
from tests.test_trace_estimators import well_posed
from flowfusion_amd import adaptive, device_adaptive, trace_estimators as TE
DEV = "cuda"
name, kind = sys.argv[1], sys.argv[2]
meta, a = load_golden(name)
kw, probes = (dict(hutchpp=True, hpp_rank=meta["hpp_rank"], hpp_vecs=meta["hpp_vecs"]), [a["S"], a["G"]]) if kind == "hutchpp" \
    else (dict(xtrace=True, xt_vecs=meta["xt_vecs"]), [a["O"]])
ok = well_posed(probes[0])
x = a["x"][ok]
cond = a["cond"][ok].to(DEV) if "cond" in a else None
pr = [p[:, ok].contiguous().to(DEV) for p in probes]
sm = score_model(meta, a, DEV, **kw)
def run():
    q = list(pr)
    TE.draw_probes = lambda n, like: q.pop(0)
    return sm.log_prob(x.to(DEV), conditional=cond)
device_adaptive.TRACE = []
ld = run()
dev_trace = device_adaptive.TRACE
device_adaptive.TRACE = None
host_trace, host_td = [], []
orig = adaptive.Dopri5._norms
def spy(self, terms, check=None):
    r = orig(self, terms, check)
    if check is not None:
        host_trace.append(r[0])
    return r
adaptive.Dopri5._norms = spy
orig_attempt = adaptive.Dopri5._attempt
def spy2(self, t0, dt, t1, *aa):
    host_td.append((t0, dt))
    return orig_attempt(self, t0, dt, t1, *aa)
adaptive.Dopri5._attempt = spy2
os.environ["FF_HOST_CONTROLLER"] = "1"
os.environ["FF_TORCH_ESTIMATOR"] = "1"       # rounds 1-3's route proper: the torch statement of the estimator
lh = run()
print("attempt | device: (attempts, accepted, t_after, dt_next, ratio) | host: (t_before, dt), ratio")
for i in range(max(len(dev_trace), len(host_trace))):
    d = dev_trace[i] if i < len(dev_trace) else None
    h = (host_td[i], host_trace[i]) if i < len(host_trace) else None
    print(i, d, h)
print("max |lp_dev - lp_host|", float((ld - lh).abs().max()), "max |lp|", float(lh.abs().max()))
