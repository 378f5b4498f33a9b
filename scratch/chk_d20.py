import sys, torch
sys.path.insert(0, "/root/repo")
from flowfusion_amd import diffusion as D
torch.manual_seed(41)
for Dm, C, units, sde in ((20, 0, [100, 200], "SUBVPSDE"), (32, 8, [256] * 4, "VESDE"), (20, 0, [100, 200], "VPSDE")):
    torch.manual_seed(41)
    sm = D.ScoreModel(D.MLP(Dm, C, 8, units), getattr(D, sde)(), no_sigma=False).eval().to("cuda")
    z = torch.randn(129, Dm, device="cuda")
    cond = torch.randn(129, C, device="cuda") if C else None
    for prec in ("f32", "bf16x2"):
        sm.precision = prec
        for method in ("bosh3", "fehlberg2", "adaptive_heun"):
            try:
                x, _ = sm.sample_ode_from_base(z, conditional=cond, method=method, atol=1e-6, rtol=1e-6)
                print(sde, Dm, prec, method, "ok", sm.last_solver_stats, float(x.abs().max()))
            except Exception as e:
                print(sde, Dm, prec, method, "FAILED", str(e)[:80])
