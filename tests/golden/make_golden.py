"""Generate the golden vectors under tests/golden/ from the reference's own code.

Run ONLY in the authoring container (needs /root/reference):

    python tests/golden/make_golden.py

The reference imports ``torchdiffeq`` at module import time (flowfusion/diffusion.py:5, flow.py:3);
that package is not installed offline.  None of the functions exercised here touches it -- they
are the pure-torch network, SDE schedules, ODE right-hand sides, divergence estimators and the
Euler-Maruyama loop -- so the two names are bound to a placeholder that raises if it is ever
called.  Everything that would go through ``odeint`` is therefore NOT covered by reference output
("parity unpinned" for the stepper); the ``hybrid_*`` cases pair the reference's right-hand sides
with the oracle's restated fixed-grid stepper instead.

Only numbers are stored (weights, inputs, outputs as .npz); no reference source is copied.
"""
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))


def _import_reference():
    def _absent(*a, **k):
        raise RuntimeError("torchdiffeq is not available offline; this code path is not pinned")
    stub = types.ModuleType("torchdiffeq")
    stub.odeint = _absent
    stub.odeint_adjoint = _absent
    sys.modules.setdefault("torchdiffeq", stub)
    sys.path.insert(0, "/root/reference")
    import flowfusion.diffusion as rd
    import flowfusion.flow as rf
    return rd, rf


def _sd(module, prefix=""):
    return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def _save(name, meta, **arrays):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    out["__meta__"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(HERE / f"{name}.npz", **out)
    print(f"wrote {name}.npz  ({sum(v.nbytes for v in out.values()) / 1024:.0f} KiB raw)")


MLP_CASES = {
    # name: (n_dimensions, n_conditionals, embedding_dimensions, units)
    "c1_2d_3x128": (2, 0, 8, [128, 128, 128]),
    "c2_16d_4x256": (16, 0, 8, [256, 256, 256, 256]),
    "cond_5d_c3_ragged": (5, 3, 6, [64, 100]),
    "c5_32d_c8_2x96": (32, 8, 8, [96, 96]),
}


def make_mlp_and_sde(rd):
    """(i) MLP.forward with scalar and vector t; (ii) SDE schedule scalars on a time grid."""
    for name, (D, C, E, units) in MLP_CASES.items():
        torch.manual_seed(100 + D)
        m = rd.MLP(n_dimensions=D, n_conditionals=C, embedding_dimensions=E, units=units)
        B = 12
        x = torch.randn(B, D)
        cond = torch.randn(B, C) if C else None
        tv = torch.rand(B)
        ts = torch.tensor(0.37)
        with torch.no_grad():
            out_v = m(tv, x, conditional=cond)
            out_s = m(ts, x, conditional=cond)
        arrays = dict(x=x, t_vec=tv, t_scalar=ts, out_vec=out_v, out_scalar=out_s, **_sd(m, "model."))
        if C:
            arrays["cond"] = cond
        _save(f"mlp_{name}", dict(D=D, C=C, E=E, units=units), **arrays)

    t = torch.cat([torch.linspace(1e-5, 1.0, 41), torch.tensor([1e-3, 0.5, 0.999])])
    x = torch.randn(t.numel(), 3)
    out = {}
    for key, sde in (("vp", rd.VPSDE()), ("ve", rd.VESDE()), ("subvp", rd.SUBVPSDE()),
                     ("vp_b", rd.VPSDE(beta_min=0.2, beta_max=12.0, T=1.0, epsilon=1e-2)),
                     ("ve_b", rd.VESDE(sigma_min=0.05, sigma_max=25.0, T=1.0, epsilon=1e-4))):
        out[f"{key}_sigma"] = sde.sigma(t)
        out[f"{key}_diffusion"] = sde.diffusion(t, x)
        out[f"{key}_drift"] = sde.drift(t, x)
        m_, s_ = sde.marginal_prob_scalars(t)
        out[f"{key}_mean_scalar"] = m_
        out[f"{key}_std_scalar"] = s_
        out[f"{key}_epsilon"] = sde.epsilon
        out[f"{key}_prior_logprob"] = sde.prior(x.shape).log_prob(x)
    _save("sde_schedules", dict(note="t grid and x are inputs; see keys"), t=t, x=x, **out)


SCORE_CASES = {
    # name: (mlp case, sde ctor name, sde kwargs, no_sigma)
    "vp_nosigma_16d": ("c2_16d_4x256", "VPSDE", {}, True),
    "vp_sigma_cond": ("cond_5d_c3_ragged", "VPSDE", {}, False),
    "ve_sigma_2d": ("c1_2d_3x128", "VESDE", {}, False),
    "ve_nosigma_cond32": ("c5_32d_c8_2x96", "VESDE", {}, True),
    "subvp_sigma_cond": ("cond_5d_c3_ragged", "SUBVPSDE", {}, False),
}


def make_score_rhs(rd):
    """(iii) ScoreModel.forward: plain RHS, Hutchinson (given e) and exact trace."""
    for name, (mlp_case, sde_name, sde_kw, no_sigma) in SCORE_CASES.items():
        D, C, E, units = MLP_CASES[mlp_case]
        torch.manual_seed(200 + D)
        m = rd.MLP(n_dimensions=D, n_conditionals=C, embedding_dimensions=E, units=units)
        sde = getattr(rd, sde_name)(**sde_kw)
        sm = rd.ScoreModel(model=m, sde=sde, no_sigma=no_sigma)
        B = 10
        x = torch.randn(B, D) * 1.5
        cond = torch.randn(B, C) if C else None
        arrays = dict(x=x, **_sd(sm))
        if C:
            arrays["cond"] = cond
        e = torch.sign(torch.randn(B, D))
        arrays["e"] = e
        for i, tval in enumerate((0.013, 0.41, 1.0)):
            t = torch.tensor(tval)
            sm.conditional = cond
            sm.prob = False
            xdot = sm.forward(t.clone(), (x.clone(),))
            arrays[f"t{i}"] = t
            arrays[f"xdot_{i}"] = xdot
            sm.prob = True
            sm.hutch = True
            sm.e = e
            xd_h, div_h = sm.forward(t.clone(), (x.clone(), torch.zeros(B, 1)))
            arrays[f"div_hutch_{i}"] = div_h
            arrays[f"xdot_hutch_{i}"] = xd_h
            sm.hutch = False
            xd_e, div_e = sm.forward(t.clone(), (x.clone(), torch.zeros(B, 1)))
            arrays[f"div_exact_{i}"] = div_e
            arrays[f"score_{i}"] = sm.score(t * torch.ones(B), x, conditional=cond)
        _save(f"score_{name}", dict(D=D, C=C, E=E, units=units, sde=sde_name, sde_kw=sde_kw, no_sigma=no_sigma), **arrays)


def make_sample_sde(rd):
    """(iv) ScoreModel.sample_sde with the RNG stream captured by replaying it."""
    for name, (mlp_case, sde_name, sde_kw, no_sigma) in SCORE_CASES.items():
        D, C, E, units = MLP_CASES[mlp_case]
        torch.manual_seed(300 + D)
        m = rd.MLP(n_dimensions=D, n_conditionals=C, embedding_dimensions=E, units=units)
        sde = getattr(rd, sde_name)(**sde_kw)
        sm = rd.ScoreModel(model=m, sde=sde, no_sigma=no_sigma)
        B, steps = 24, 7
        cond = torch.randn(B, C) if C else None
        torch.manual_seed(4242)
        out = sm.sample_sde((B, D), conditional=cond, steps=steps)
        # replay the draws: one prior sample, then one randn_like per step (diffusion.py:532-536, 554)
        torch.manual_seed(4242)
        x_prior = sde.prior([D]).sample([B])
        noise = torch.stack([torch.randn_like(x_prior) for _ in range(steps)])
        arrays = dict(x_prior=x_prior, noise=noise, out=out, **_sd(sm))
        if C:
            arrays["cond"] = cond
        _save(f"sde_{name}", dict(D=D, C=C, E=E, units=units, sde=sde_name, sde_kw=sde_kw, no_sigma=no_sigma,
                                  steps=steps, seed=4242), **arrays)


FLOW_CASES = {
    "flow_3d": dict(target_dimension=3, hidden_units=[64, 64]),
    "flow_16d_ragged": dict(target_dimension=16, hidden_units=[128, 96, 128]),
    "cflow_4d_c2": dict(target_dimension=4, conditional_dimension=2, hidden_units=[64, 64]),
    "cflow_8d_c5": dict(target_dimension=8, conditional_dimension=5, hidden_units=[128, 128]),
}


def _make_flow(rf, kw, seed):
    torch.manual_seed(seed)
    D = kw["target_dimension"]
    extra = dict(target_shift=torch.randn(D), target_scale=torch.rand(D) + 0.5)
    if "conditional_dimension" in kw:
        C = kw["conditional_dimension"]
        extra.update(conditional_shift=torch.randn(C), conditional_scale=torch.rand(C) + 0.5)
        return rf.ConditionalODEFlow(**kw, **extra)
    return rf.ODEFlow(**kw, **extra)


def make_flow(rf):
    """(v) flow dynamics and dynamics_with_jacobian with non-trivial shift/scale."""
    for i, (name, kw) in enumerate(FLOW_CASES.items()):
        f = _make_flow(rf, kw, 500 + i)
        D = kw["target_dimension"]
        C = kw.get("conditional_dimension", 0)
        B = 9
        x = torch.randn(B, D)
        cond = torch.randn(B, C) * 2 + 1 if C else None
        arrays = dict(x=x, **_sd(f))
        if C:
            arrays["cond"] = cond
        for j, tval in enumerate((0.0, 0.3, 1.0)):
            t = torch.tensor(tval)
            arrays[f"t{j}"] = t
            with torch.no_grad():
                arrays[f"v_{j}"] = f.dynamics(t, (x, cond))[0] if C else f.dynamics(t, (x,))
            states = (x.clone(), cond, torch.zeros(B, 1)) if C else (x.clone(), torch.zeros(B, 1))
            res = f.dynamics_with_jacobian(t.clone(), states)
            arrays[f"div_{j}"] = res[-1]
        _save(name, dict(kw=kw), **arrays)


def make_hybrid(rd, rf):
    """(vi) the oracle's restated fixed-grid stepper driving the REFERENCE's right-hand sides."""
    from oracle import flowfusion_oracle as O

    for name, (mlp_case, sde_name, sde_kw, no_sigma) in SCORE_CASES.items():
        D, C, E, units = MLP_CASES[mlp_case]
        torch.manual_seed(700 + D)
        m = rd.MLP(n_dimensions=D, n_conditionals=C, embedding_dimensions=E, units=units)
        sde = getattr(rd, sde_name)(**sde_kw)
        sm = rd.ScoreModel(model=m, sde=sde, no_sigma=no_sigma, hutchinson=True)
        B = 16
        base = torch.randn(B, D)
        cond = torch.randn(B, C) if C else None
        eps = float(sde.epsilon)
        arrays = dict(base=base, **_sd(sm))
        if C:
            arrays["cond"] = cond
        meta = dict(D=D, C=C, E=E, units=units, sde=sde_name, sde_kw=sde_kw, no_sigma=no_sigma, runs=[])
        for method, nsteps in (("rk4", 12), ("euler", 25), ("midpoint", 10)):
            opts = {"step_size": (1.0 - eps) / nsteps}
            # sampling: t 1 -> eps  (reference call site diffusion.py:605-639)
            z = base * sde.sigma_max if hasattr(sde, "sigma_max") else base
            sm.prob = False
            sm.conditional = cond
            func = lambda t, y: (sm.forward(t, (y[0],)).detach(),)
            (x0,) = O.odeint_fixed(func, (z,), torch.tensor([1.0, sde.epsilon]), method, opts)
            arrays[f"sample_{method}"] = x0
            # log-density: t eps -> 1 with Hutchinson probe (diffusion.py:697-752, 814)
            x_data = torch.randn(B, D) * 0.7
            e = torch.sign(torch.randn(B, D))
            sm.prob = True
            sm.hutch = True
            sm.e = e
            func2 = lambda t, y: tuple(v.detach() for v in sm.forward(t, (y[0].detach(), y[1])))
            xT, dlp = O.odeint_fixed(func2, (x_data, torch.zeros(B, 1)), torch.tensor([sde.epsilon, 1.0]), method, opts)
            lp = dlp + torch.sum(sde.prior(xT.shape).log_prob(xT), dim=1, keepdim=True)
            sm.hutch = False
            func3 = lambda t, y: tuple(v.detach() for v in sm.forward(t, (y[0].detach(), y[1])))
            xT_e, dlp_e = O.odeint_fixed(func3, (x_data, torch.zeros(B, 1)), torch.tensor([sde.epsilon, 1.0]), method, opts)
            lp_e = dlp_e + torch.sum(sde.prior(xT_e.shape).log_prob(xT_e), dim=1, keepdim=True)
            arrays.update({f"x_data_{method}": x_data, f"e_{method}": e, f"xT_{method}": xT,
                           f"lp_hutch_{method}": lp, f"lp_exact_{method}": lp_e})
            meta["runs"].append(dict(method=method, nsteps=nsteps, step_size=opts["step_size"]))
        _save(f"hybrid_score_{name}", meta, **arrays)

    for i, (name, kw) in enumerate(FLOW_CASES.items()):
        f = _make_flow(rf, kw, 800 + i)
        D = kw["target_dimension"]
        C = kw.get("conditional_dimension", 0)
        B = 16
        xT = torch.randn(B, D)
        cond = torch.randn(B, C) * 2 + 1 if C else None
        arrays = dict(xT=xT, **_sd(f))
        if C:
            arrays["cond"] = cond
        meta = dict(kw=kw, runs=[])
        for method, nsteps in (("rk4", 10), ("euler", 20)):
            opts = {"step_size": 1.0 / nsteps}
            with torch.no_grad():
                if C:
                    func = lambda t, y: (f.dynamics(t, (y[0], cond))[0],)
                else:
                    func = lambda t, y: (f.dynamics(t, (y[0],)),)
                (x0,) = O.odeint_fixed(func, (xT,), torch.tensor([1.0, 0.0]), method, opts)
                sample = x0 * f.target_scale + f.target_shift            # flow.py:303-305
            x_data = torch.randn(B, D) * f.target_scale + f.target_shift
            xn = (x_data - f.target_shift) / f.target_scale               # flow.py:421
            if C:
                func2 = lambda t, y: tuple(v.detach() for v in
                                           (lambda r: (r[0], r[2]))(f.dynamics_with_jacobian(t, (y[0].detach(), cond, y[1]))))
            else:
                func2 = lambda t, y: tuple(v.detach() for v in f.dynamics_with_jacobian(t, (y[0].detach(), y[1])))
            xT2, logj = O.odeint_fixed(func2, (xn, torch.zeros(B, 1)), torch.tensor([0.0, 1.0]), method, opts)
            lp = torch.sum(-0.5 * xT2 ** 2 - 0.5 * torch.log(f.twopi), dim=1) + logj.squeeze(1) \
                - torch.sum(torch.log(f.target_scale))                    # flow.py:434-438
            arrays.update({f"sample_{method}": sample, f"x_data_{method}": x_data, f"logprob_{method}": lp})
            meta["runs"].append(dict(method=method, nsteps=nsteps, step_size=opts["step_size"]))
        _save(f"hybrid_{name}", meta, **arrays)


TRACE_CASES = {
    # name: (score case, hpp_rank, hpp_vecs, xt_vecs)
    "vp_sigma_cond": ("vp_sigma_cond", 2, 3, 3),
    "ve_nosigma_cond32": ("ve_nosigma_cond32", 3, 2, 4),
    "vp_nosigma_16d": ("vp_nosigma_16d", 1, 1, 2),
}


def make_trace_estimators(rd):
    """Hutch++ and XTrace (diffusion.py:336-481): the reference's ScoreModel.forward with stored probes S, G, O
    at three times, and hybrid log-densities (the oracle's fixed-grid stepper driving that forward)."""
    from oracle import flowfusion_oracle as O

    for name, (score_case, r, mv, xt) in TRACE_CASES.items():
        mlp_case, sde_name, sde_kw, no_sigma = SCORE_CASES[score_case]
        D, C, E, units = MLP_CASES[mlp_case]
        torch.manual_seed(900 + D)
        m = rd.MLP(n_dimensions=D, n_conditionals=C, embedding_dimensions=E, units=units)
        sde = getattr(rd, sde_name)(**sde_kw)
        sm = rd.ScoreModel(model=m, sde=sde, no_sigma=no_sigma, hpp_rank=r, hpp_vecs=mv, xt_vecs=xt)
        B = 10
        x = torch.randn(B, D) * 1.2
        cond = torch.randn(B, C) if C else None
        S = torch.sign(torch.randn(r, B, D))
        G = torch.sign(torch.randn(mv, B, D))
        Om = torch.sign(torch.randn(xt, B, D))
        arrays = dict(x=x, S=S, G=G, O=Om, **_sd(sm))
        if C:
            arrays["cond"] = cond
        sm.conditional = cond
        sm.prob = True
        sm.hutch = False
        sm.S, sm.G, sm.O = S, G, Om

        def rhs(kind):
            sm.hutchpp, sm.xtrace = kind == "hpp", kind == "xt"
            return lambda t, y: tuple(v.detach() for v in sm.forward(t, (y[0].detach(), y[1])))

        for i, tval in enumerate((0.013, 0.41, 1.0)):
            t = torch.tensor(tval)
            arrays[f"t{i}"] = t
            for kind in ("hpp", "xt", "exact"):
                xd, div = rhs(kind)(t.clone(), (x.clone(), torch.zeros(B, 1)))
                arrays[f"div_{kind}_{i}"] = div
            arrays[f"xdot_{i}"] = xd
        eps = float(sde.epsilon)
        opts = {"step_size": (1.0 - eps) / 8}
        for kind in ("hpp", "xt"):
            xT, dlp = O.odeint_fixed(rhs(kind), (x, torch.zeros(B, 1)), torch.tensor([sde.epsilon, 1.0]), "rk4", opts)
            arrays[f"lp_{kind}_rk4"] = dlp + torch.sum(sde.prior(xT.shape).log_prob(xT), dim=1, keepdim=True)
        _save(f"trace_{name}", dict(D=D, C=C, E=E, units=units, sde=sde_name, sde_kw=sde_kw, no_sigma=no_sigma,
                                    hpp_rank=r, hpp_vecs=mv, xt_vecs=xt, step_size=opts["step_size"]), **arrays)


def make_population(rd):
    """PopulationModelDiffusion[Conditional] (diffusion.py:1466-1848): state_dict layout and ``sample_sde`` with
    the random stream captured (the wrappers always take 100 steps, whatever ``steps`` says)."""
    for name, C in (("pop_4d", 0), ("popcond_4d_c2", 2)):
        D = 4
        torch.manual_seed(950 + C)
        m = rd.MLP(n_dimensions=D, n_conditionals=C, embedding_dimensions=6, units=[64, 48])
        shift, scale = torch.randn(D), torch.rand(D) + 0.5
        if C:
            cshift, cscale = torch.randn(C), torch.rand(C) + 0.5
            pm = rd.PopulationModelDiffusionConditional(model=m, sde=rd.VESDE(), shift=shift, scale=scale,
                                                        conditional_shift=cshift, conditional_scale=cscale)
        else:
            pm = rd.PopulationModelDiffusion(model=m, sde=rd.VESDE(), shift=shift, scale=scale)
        B = 12
        cond = torch.randn(B, C) * 2 + 1 if C else None
        torch.manual_seed(777)
        out = pm.sample_sde((B, D), cond, steps=7) if C else pm.sample_sde((B, D), steps=7)
        torch.manual_seed(777)
        x_prior = pm.sde.prior([D]).sample([B])
        noise = torch.stack([torch.randn_like(x_prior) for _ in range(100)])
        arrays = dict(x_prior=x_prior, noise=noise, out=out, **_sd(pm))
        if C:
            arrays["cond"] = cond
        _save(name, dict(D=D, C=C, E=6, units=[64, 48], seed=777), **arrays)


if __name__ == "__main__":
    rd, rf = _import_reference()
    torch.set_num_threads(4)
    only = set(sys.argv[1:])          # e.g. `make_golden.py trace` regenerates one family
    if not only or "mlp" in only:
        make_mlp_and_sde(rd)
    if not only or "score" in only:
        make_score_rhs(rd)
    if not only or "sde" in only:
        make_sample_sde(rd)
    if not only or "flow" in only:
        make_flow(rf)
    if not only or "hybrid" in only:
        make_hybrid(rd, rf)
    if not only or "trace" in only:
        make_trace_estimators(rd)
    if not only or "pop" in only:
        make_population(rd)
