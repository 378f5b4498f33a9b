"""End-to-end accuracy of 16-bit-operand arithmetic (two round-to-nearest bf16 parts, three products hh + hm + mh, fp32
accumulation) for BASELINE config 2 / 3 (16-dim VP, 4x256, 100-step RK4): the CPU oracle with every Linear layer's
product replaced by the split arithmetic, against the float64 oracle.  Planning data for a `precision="bf16x2"` option."""
import sys
import torch
sys.path.insert(0, ".")
from oracle import flowfusion_oracle as O
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel

torch.manual_seed(0)
sm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True).eval()
sd = {k: v.detach() for k, v in sm.state_dict().items()}
so64 = O.ScoreOracle(O.mlp_params_from_state_dict(sd, "model."), O.VP(dtype=torch.float64), no_sigma=True, dtype=torch.float64)
so32 = O.ScoreOracle(O.mlp_params_from_state_dict(sd, "model."), O.VP(), no_sigma=True)
eps = float(sm.sde.epsilon)
opts = {"step_size": (1.0 - eps) / 100}
B = 192
z = torch.randn(B, 16)
xq = torch.randn(64, 16) * 0.9
e = torch.sign(torch.randn(64, 16))
ref = so64.sample_ode_from_base(z.double(), None, "rk4", opts)
lref = so64.log_prob(xq.double(), None, "rk4", opts, "hutch", e.double())
real_linear = torch.nn.functional.linear


def parts(t, n, rne):
    out, r = [], t
    for _ in range(n):
        if rne:
            p = r.bfloat16().float()
        else:
            p = (r.contiguous().view(torch.int32) & torch.tensor(-65536, dtype=torch.int32)).view(torch.float32)
        out.append(p)
        r = r - p
    return out


def make_linear(n, rne, pairs):
    def lin(x, w, b=None):
        if x.dtype != torch.float32:
            return real_linear(x, w, b)
        xs, ws = parts(x, n, rne), parts(w, n, rne)
        y = sum(real_linear(xs[i], ws[j]) for i, j in pairs)
        return y if b is None else y + b
    return lin


def run(name, lin):
    torch.nn.functional.linear = lin
    try:
        x = so32.sample_ode_from_base(z, None, "rk4", opts)
        lp = so32.log_prob(xq, None, "rk4", opts, "hutch", e)
    finally:
        torch.nn.functional.linear = real_linear
    se = float((x.double() - ref).abs().max() / ref.abs().max())
    le = float(((lp.double() - lref).abs() / lref.abs().clamp_min(1.0)).max())
    print(f"{name:58s} state err / max|x| {se:.2e}   log_prob rel err {le:.2e}")


run("fp32 (the oracle as it is)", real_linear)
run("bf16 x 3 truncation, 6 products (the shipped bf16x3)", make_linear(3, False, [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]))
run("bf16 x 2 round-to-nearest, 3 products (hh hm mh)", make_linear(2, True, [(0, 0), (0, 1), (1, 0)]))
run("bf16 x 2 round-to-nearest, 4 products (+ mm)", make_linear(2, True, [(0, 0), (0, 1), (1, 0), (1, 1)]))
run("bf16 x 2 truncation, 3 products", make_linear(2, False, [(0, 0), (0, 1), (1, 0)]))
run("bf16 x 3 round-to-nearest, 5 products (no mm)", make_linear(3, True, [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0)]))
run("plain bf16 (1 product)", make_linear(1, True, [(0, 0)]))
