import torch, sys, os
sys.path.insert(0, '/root/repo')
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel
from oracle import flowfusion_oracle as O
from flowfusion_amd import _native
torch.manual_seed(0)
for units in ([256]*4, [256], [256]*2, [128]*4):
    sm = ScoreModel(MLP(16, 0, 8, units), VPSDE(), no_sigma=True).eval()
    sd = {k: v.detach().clone() for k, v in sm.state_dict().items()}
    so = O.ScoreOracle(O.mlp_params_from_state_dict(sd, "model."), O.VP(), no_sigma=True)
    sm = sm.to('cuda')
    p = sm._net().plan(0)
    z = torch.randn(256, 16)
    for n in (1, 4, 25, 100):
        opts = {"step_size": (1 - 1e-3) / n}
        x, _ = sm.sample_ode_from_base(z.cuda(), method="rk4", options=opts)
        ref = so.sample_ode_from_base(z, None, "rk4", opts)
        err = ((x.cpu() - ref).abs().max() / ref.abs().max()).item()
        print(f"units={units} kernel={_native.lib().ff_kernel_name(p.kernel_id).decode()} steps={n}: err {err:.2e}", flush=True)
