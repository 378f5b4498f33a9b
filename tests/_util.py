"""Helpers shared by the tests: golden-fixture loading and oracle / product object construction."""
import json
from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"


def load_golden(name):
    z = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    meta = json.loads(bytes(z["__meta__"]).decode())
    arrays = {k: torch.from_numpy(np.array(z[k])) for k in z.files if k != "__meta__"}
    return meta, arrays


def golden_names(prefix):
    return sorted(p.stem for p in GOLDEN.glob(f"{prefix}*.npz"))


def sde_oracle(name, kw, dtype=torch.float32):
    from oracle import flowfusion_oracle as O
    return {"VPSDE": O.VP, "VESDE": O.VE, "SUBVPSDE": O.SubVP}[name](**kw, dtype=dtype)


def score_oracle(meta, arrays, dtype=torch.float32):
    from oracle import flowfusion_oracle as O
    p = O.mlp_params_from_state_dict(arrays, "model.")
    return O.ScoreOracle(p, sde_oracle(meta["sde"], meta["sde_kw"], dtype), no_sigma=meta["no_sigma"], dtype=dtype)


def score_model(meta, arrays, device="cpu", **kw):
    """Product ScoreModel loaded from a reference state_dict stored in a fixture."""
    from flowfusion_amd import diffusion as D
    m = D.MLP(n_dimensions=meta["D"], n_conditionals=meta["C"], embedding_dimensions=meta["E"], units=meta["units"])
    sde = getattr(D, meta["sde"])(**meta["sde_kw"])
    sm = D.ScoreModel(model=m, sde=sde, no_sigma=meta["no_sigma"], **kw)
    sd = {k: v for k, v in arrays.items() if k.startswith("model.") or k.startswith("sde.")}
    sm.load_state_dict(sd, strict=True)
    return sm.to(device).eval()


def flow_oracle(arrays, dtype=torch.float32):
    from oracle import flowfusion_oracle as O
    return O.FlowOracle(O.flow_params_from_state_dict(arrays), dtype=dtype)


def flow_model(meta, arrays, device="cpu"):
    from flowfusion_amd import flow as F
    kw = meta["kw"]
    cls = F.ConditionalODEFlow if "conditional_dimension" in kw else F.ODEFlow
    f = cls(**kw)
    keys = set(f.state_dict().keys())
    f.load_state_dict({k: v for k, v in arrays.items() if k in keys}, strict=True)
    return f.to(device).eval()


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def max_rel(a, b, floor=1.0):
    """max_i |a_i - b_i| / max(|b_i|, floor): elementwise relative error with an absolute floor."""
    a, b = a.double(), b.double()
    return ((a - b).abs() / b.abs().clamp_min(floor)).max().item()
