import torch, time, sys
sys.path.insert(0, '/root/repo')
from flowfusion_amd.diffusion import MLP, VPSDE, VESDE, ScoreModel
from flowfusion_amd import solvers
torch.manual_seed(0)
dev = 'cuda'
def rk_ref(sm, z, t_span, method, options, cond=None):
    # independent torch loop with the same grid rules (float64 for a tight reference)
    plan = solvers.plan_ode(t_span, method, options)
    tab = solvers.FIXED_METHODS[method]
    S = tab.stages
    x = z.double()
    smd = sm
    k = [None]*S
    for e in range(plan.t_eval.numel()):
        i = e % S
        y = x.clone()
        for j in range(i):
            y = y + plan.cin[e, j].double().to(dev) * k[j]
        t = plan.t_eval[e].to(dev)
        with torch.no_grad():
            f = plan.sign * smd.ode_drift(t.double(), y, conditional=cond)
        k[i] = f
        if i == S-1:
            for j in range(S):
                x = x + plan.cout[e, j].double().to(dev) * k[j]
    return x

for (D, C, units, sde, no_sigma) in [(16, 0, [256]*4, VPSDE(), True), (2, 0, [128]*3, VESDE(), False), (32, 8, [256]*4, VESDE(), False), (5, 3, [64, 100], VPSDE(), False)]:
    m = MLP(D, C, 8, units).to(dev)
    sm = ScoreModel(m, sde.to(dev), no_sigma=no_sigma).eval()
    B = 1000
    z = torch.randn(B, D, device=dev)
    cond = torch.randn(B, C, device=dev) if C else None
    eps = float(sde.epsilon)
    opts = {'step_size': (1-eps)/20}
    x, _ = sm.sample_ode_from_base(z, conditional=cond, method='rk4', options=opts)
    torch.cuda.synchronize()
    import copy
    smd = copy.deepcopy(sm).double()
    zz = z * sde.sigma_max if hasattr(sde, 'sigma_max') else z
    xr = rk_ref(smd, zz, torch.tensor([1.0, eps]), 'rk4', opts, cond=None if cond is None else cond.double())
    err = (x.double() - xr).abs().max().item()
    print(f"D={D} C={C} units={units} {type(sde).__name__}: max|x| {xr.abs().max().item():.3f} max abs err {err:.3e}")
