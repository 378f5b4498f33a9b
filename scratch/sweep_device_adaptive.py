"""Randomised sweep (round 3): default-style adaptive calls under the device and the host step controller over random
shapes / SDEs / modes / tolerances / methods / options -- results must agree within what two adaptive solves can (2e-4, or both
raise the same torchdiffeq assertion) and attempt counts must be close.  Not part of the test suite (minutes of GPU time)."""
import os, random, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowfusion_amd import diffusion as D, flow as F
dev = torch.device("cuda", 0)
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
def run(fn):
    out = []
    for env in (None, "1"):
        if env: os.environ["FF_HOST_CONTROLLER"] = env
        else: os.environ.pop("FF_HOST_CONTROLLER", None)
        try:
            torch.manual_seed(7)
            r, st = fn()
            out.append((r.detach().float().cpu(), st))
        except RuntimeError as e:
            out.append((str(e), getattr(e, "solver_stats", None)))
    os.environ.pop("FF_HOST_CONTROLLER", None)
    return out
for case in range(N):
    kind = rnd.choice(["score", "score", "flow", "cflow"])
    Dm = rnd.choice([1, 2, 3, 5, 8, 16, 20, 33])
    C = rnd.choice([0, 0, 2, 7]) if kind != "flow" else 0
    if kind == "cflow" and C == 0: C = 3
    units = [rnd.choice([24, 64, 100, 128, 200, 256]) for _ in range(rnd.choice([1, 2, 3, 4]))]
    B = rnd.choice([1, 5, 64, 300, 1000, 4099])
    tol = rnd.choice([1e-3, 1e-4, 1e-5, 1e-6])
    method = rnd.choice(["dopri5", "dopri5", "dopri5", "bosh3", "fehlberg2", "adaptive_heun"])
    if method != "dopri5": tol = max(tol, 1e-4)
    opts = rnd.choice([None, None, {"min_step": 1e-6}, {"max_step": 0.05}, {"first_step": 0.01}, {"max_num_steps": 500}])
    what = rnd.choice(["sample", "logp_exact", "logp_hutch"])
    torch.manual_seed(1000 + case)
    x = torch.randn(B, Dm, device=dev)
    cond = torch.randn(B, C, device=dev) if C else None
    if kind == "score":
        sde = rnd.choice(["VESDE", "VESDE", "VPSDE", "SUBVPSDE"])
        ns = rnd.random() < 0.5
        m = D.ScoreModel(D.MLP(Dm, C, rnd.choice([4, 8, 16]), units), getattr(D, sde)(), no_sigma=ns).eval().to(dev)
        tag = (case, kind, sde, ns, Dm, C, units, B, tol, method, opts, what)
        if what == "sample":
            fn = lambda: (m.sample_ode_from_base(x, conditional=cond, atol=tol, rtol=tol, method=method, options=opts)[0], dict(m.last_solver_stats))
        else:
            m.hutch = what == "logp_hutch"
            o2 = opts if opts is not None else {"min_step": 1e-6}
            fn = lambda: (m.log_prob(x * 0.5, conditional=cond, atol=tol, rtol=tol, method=method, options=o2), dict(m.last_solver_stats))
    else:
        m = (F.ODEFlow(Dm, units) if kind == "flow" else F.ConditionalODEFlow(Dm, C, units)).eval().to(dev)
        args = () if kind == "flow" else (cond,)
        tag = (case, kind, Dm, C, units, B, tol, method, opts, what)
        if what == "sample":
            fn = lambda: (m.sample(x, *args, method=method, options=opts, atol=tol, rtol=tol), dict(m.last_solver_stats))
        else:
            fn = lambda: (m.log_prob(x * 0.5, *args, atol=tol, rtol=tol, method=method, options=opts, hutchinson=(what == "logp_hutch")), dict(m.last_solver_stats))
    (rd, sd), (rh, sh) = run(fn)
    if isinstance(rd, str) or isinstance(rh, str):
        ok = isinstance(rd, str) and isinstance(rh, str) and rd.split(" dt ")[0] == rh.split(" dt ")[0]
        msg = f"raise: dev={rd!r} host={rh!r} stats {sd} {sh}"
    else:
        err = float(((rd - rh).abs() / rh.abs().clamp_min(max(1.0, float(rh.abs().max()) if what == "sample" else 1.0))).max())
        close = abs(sd["attempts"] - sh["attempts"]) <= max(3, 0.1 * sh["attempts"])
        ok = err < max(2e-4, 30 * tol) and close and bool(torch.isfinite(rd).all())
        msg = f"err {err:.2e} attempts {sd['attempts']}/{sh['attempts']} accepted {sd['accepted']}/{sh['accepted']} chunks {sd.get('chunks')}"
    bad += not ok
    print(("ok  " if ok else "FAIL"), tag, msg, flush=True)
print("failures:", bad)
