"""One-off diagnosis of a case of test_random_shapes_against_oracle: replay the sweep's random sequence up to case N of seed S
and print the product's and the fp32 oracle's distance from the float64 oracle (python -m tests.sweep_diag S N)."""
import random
import sys

import torch

from flowfusion_amd import diffusion as Dm
from oracle import flowfusion_oracle as O

DEV = torch.device("cuda", 0)


def main(seed, target):
    rnd = random.Random(seed)
    for case in range(target + 1):
        wmax = rnd.choice([40, 64, 100, 128, 200, 256, 384, 512])
        depth = rnd.choice([1, 2, 3, 5])
        units = [rnd.randint(max(8, wmax // 2), wmax) for _ in range(depth)]
        units[rnd.randrange(depth)] = wmax
        D = rnd.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 32, 33, 48, 64])
        C = rnd.choice([0, 0, 1, 4, 9, 16, 17, 32])
        sde = rnd.choice(["VPSDE", "VESDE", "SUBVPSDE"])
        no_sigma = rnd.random() < 0.5
        method, nsteps = rnd.choice([("euler", 12), ("midpoint", 8), ("rk4", 6), ("dopri5_fixed", 4), ("heun3", 6)])
        B = rnd.choice([1, 5, 16, 33, 100])
        emb = rnd.choice([2, 8, 10])
        mode = rnd.choice(["hutch", "exact"])
        if case < target:
            continue
        torch.manual_seed(1000 + case)
        sm = Dm.ScoreModel(Dm.MLP(D, C, emb, units), getattr(Dm, sde)(), no_sigma=no_sigma).eval()
        params = O.mlp_params_from_state_dict({k: v.detach().clone() for k, v in sm.state_dict().items()})
        mk = lambda dt: O.ScoreOracle(params, {"VPSDE": O.VP, "VESDE": O.VE, "SUBVPSDE": O.SubVP}[sde](dtype=dt), no_sigma=no_sigma, dtype=dt)
        so64, so32 = mk(torch.float64), mk(torch.float32)
        sm = sm.to(DEV)
        z = torch.randn(B, D)
        cond = torch.randn(B, C) if C else None
        cd = None if cond is None else cond.to(DEV)
        opts = {"step_size": (1.0 - float(sm.sde.epsilon)) / nsteps}
        sm.sample_ode_from_base(z.to(DEV), conditional=cd, method=method, options=opts)
        sm.hutch = mode == "hutch"
        xd = torch.randn(B, D) * 0.5
        lp = sm.log_prob(xd.to(DEV), conditional=cd, method=method, options=opts).cpu()
        e = sm.e.cpu() if sm.hutch else None
        r64 = so64.log_prob(xd.double(), None if cond is None else cond.double(), method, opts, mode, None if e is None else e.double())
        r32 = so32.log_prob(xd, cond, method, opts, mode, e)
        xT64, d64 = so64.solve_odes_forward(xd.double(), None if cond is None else cond.double(), method, opts, mode, None if e is None else e.double())
        err = lambda a: ((a.double() - r64).abs() / r64.abs().clamp_min(1.0))
        i = int(err(lp).argmax())
        print((case, D, C, units, sde, no_sigma, method, nsteps, B, mode))
        print("product vs f64 oracle:", float(err(lp).max()), " f32 oracle vs f64 oracle:", float(err(r32).max()))
        print("worst row", i, "log_prob", float(r64[i]), "delta_logp", float(d64[i]), "|xT|max", float(xT64[i].abs().max()))


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]))
