"""CPU emulation of the mlp_ode kernel's *semantics* (TEST INFRASTRUCTURE).

Reads exactly what the kernel reads -- the packed weight buffer produced by ff_mlp_wpack and the
evaluation table produced by flowfusion_amd.solvers -- and performs the same stage bookkeeping in
float64 torch ops.  It lets the CPU-only test tier check all host-side logic (packing order, table
words, time reversal signs, stage slots, noise rows, tangent columns) against the oracle without a
GPU.  The lane-level MFMA mapping itself is only exercised by the `-m gpu` tests.
"""
import torch


def feat_of_reg(tile, r, q):
    if tile == 32:
        return 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * q
    return 32 * (r >> 3) + 16 * ((r & 7) >> 2) + 4 * q + (r & 3)


def _geom(KR, NOB, gb_max):
    G = KR // 4
    GB = min(G, gb_max)
    GA = G - GB
    NC = G * NOB
    return dict(G=G, NOB=NOB, GB=GB, GA=GA, NC=NC, CPAD=(NC + 15) // 16 * 16)


def _chunk_gob(L, c):
    na = L["GA"] * L["NOB"]
    if c < na:
        return c // L["NOB"], c % L["NOB"]
    return L["GA"] + (c - na) % L["GB"], (c - na) // L["GB"]


def decode_wpack(plan, wpack):
    """Invert the packing (csrc/ff_layout.h): dense zero-padded matrices W1[H, dx + dc] (dx/dc =
    feature capacity of the state / conditional registers), hidden [H,H] + bias[H], output
    [nob_out*32, H] + bias."""
    D, C, NH, H, dregs, cregs, tile = plan[0], plan[1], plan[2], plan[3], plan[4], plan[5], plan[7]
    NQ, PHYS, RB = 64 // tile, 32 // tile, 32 // (64 // tile)
    NB = H // 32
    nob_out = (dregs + RB - 1) // RB
    CF = 256 * PHYS
    w = wpack.double()
    g1, gh, go = _geom(dregs + cregs, NB, RB // 4), _geom(H // NQ, NB, RB // 4), _geom(H // NQ, nob_out, RB // 4)
    stream = g1["CPAD"] + (NH - 1) * gh["CPAD"] + go["CPAD"]
    dx, dc = dregs * NQ, cregs * NQ

    def decode(L, chunk0, kdim, kmap):
        M = torch.zeros(L["NOB"] * 32, kdim, dtype=torch.float64)
        blk = w[chunk0 * CF:(chunk0 + L["CPAD"]) * CF].view(L["CPAD"], PHYS, 64, 4)
        assert blk[L["NC"]:].abs().sum() == 0                    # padding chunks are zero
        for c in range(L["NC"]):
            g, ob = _chunk_gob(L, c)
            for p in range(PHYS):
                for j in range(4):
                    for q in range(NQ):
                        k = kmap(4 * g + j, q)
                        M[ob * 32 + tile * p: ob * 32 + tile * (p + 1), k] = blk[c, p, tile * q: tile * (q + 1), j]
        return M

    W1 = decode(g1, 0, dx + dc,
                lambda r, q: feat_of_reg(tile, r, q) if r < dregs else dx + feat_of_reg(tile, r - dregs, q))
    hidden = []
    bias0 = stream * CF
    for l in range(NH - 1):
        Wl = decode(gh, g1["CPAD"] + l * gh["CPAD"], H, lambda r, q: feat_of_reg(tile, r, q))
        hidden.append((Wl, w[bias0 + l * H: bias0 + (l + 1) * H]))
    Wo = decode(go, g1["CPAD"] + (NH - 1) * gh["CPAD"], H, lambda r, q: feat_of_reg(tile, r, q))
    bo = w[bias0 + (NH - 1) * H: bias0 + (NH - 1) * H + nob_out * 32]
    assert bias0 + (NH - 1) * H + nob_out * 32 == w.numel()
    return W1, hidden, Wo, bo, dx


def _act_fn(plan):
    """(value, slope) of the plan's activation (plan words 8..10: FF_ACT_* code and two float parameters),
    from torch's own definitions in float64."""
    import struct
    F = torch.nn.functional
    kind = int(plan[8]) if len(plan) > 8 else 0
    p0, p1 = (struct.unpack("f", struct.pack("i", int(plan[i])))[0] if len(plan) > i else 0.0 for i in (9, 10))
    fn = {0: F.silu, 1: torch.tanh, 2: torch.sigmoid, 3: F.relu, 4: lambda a: F.leaky_relu(a, p0),
          5: lambda a: F.elu(a, p0), 6: lambda a: F.softplus(a, p0, p1), 7: F.gelu,
          8: lambda a: F.gelu(a, approximate="tanh")}[kind]

    def act(a):
        with torch.enable_grad():
            a = a.detach().requires_grad_(True)
            h = fn(a)
            d, = torch.autograd.grad(h.sum(), a)
        return h.detach(), d

    return act


def _run(plan, wpack, etab, n_rows, x, cond, V, noise, ks, kl, lp, jac=None, jac_first=0):
    """The kernel's evaluation loop over the first n_rows rows of etab (float64).  ``jac`` ([B, D, D]): rows
    jac_first.. of the last evaluation's transposed Jacobian are stored there (ff_ode_args.jac_out)."""
    D, C, NH, H = plan[0], plan[1], plan[2], plan[3]
    W1, hidden, Wo, bo, dx = decode_wpack(plan, wpack)
    W1x, W1c = W1[:, :D], W1[:, dx:dx + C]
    _silu = _act_fn(plan)
    etab64 = etab.double()
    ints = etab.contiguous().view(torch.int32)
    cc = cond.double() @ W1c.T if C else 0.0
    for e in range(n_rows):
        a, b, gn = etab64[e, 0], etab64[e, 1], etab64[e, 2]
        flags, slot, nidx = int(ints[e, 3]), int(ints[e, 4]), int(ints[e, 5])
        cin, cout = etab64[e, 8:15], etab64[e, 16:23]
        c1 = etab64[e, 32:32 + H]
        y = x + torch.einsum("s,sbd->bd", cin, ks)
        pre = y @ W1x.T + cc + c1
        h, dh_fac = _silu(pre)
        if V is not None:
            dh = dh_fac[:, None, :] * (V @ W1x.T)               # [B,T,H]
        for Wl, bl in hidden:
            pre = h @ Wl.T + bl
            if V is not None:
                dpre = dh @ Wl.T
            h, dh_fac = _silu(pre)
            if V is not None:
                dh = dh_fac[:, None, :] * dpre
        net = (h @ Wo.T + bo)[:, :D]
        ks[slot] = a * y + b * net
        if V is not None:
            dnet = (dh @ Wo.T)[:, :, :D]                        # [B,T,D]
            kl[slot] = a * (V * V).sum((1, 2)) + b * (V * dnet).sum((1, 2))
            if jac is not None and e == n_rows - 1:
                jac[:, jac_first:jac_first + V.shape[1], :] = (a * V + b * dnet).to(jac.dtype)
        if flags & 1:
            x = x + torch.einsum("s,sbd->bd", cout, ks)
            lp = lp + torch.einsum("s,sb->b", cout, kl)
        if flags & 2:
            x = x + gn * noise[nidx].double()
    return x, lp


def _tangents(plan, mode, probe, B, first=0, count=0):
    D = plan[0]
    if mode == 0:
        return None
    if mode == 1:
        return probe.double()[:, None, :]                       # [B, T=1, D]
    eye = torch.eye(D, dtype=torch.float64)
    if count:
        eye = eye[first:first + count]
    return eye[None].expand(B, eye.shape[0], D)


def emulate(plan, wpack, etab, x_in, cond=None, probe=None, noise=None, mode=0,
            in_shift=None, in_scale=None, out_scale=None, out_shift=None, rng=None):
    """Returns (x_out [B,D], dlogp [B]) in float64.  ``rng = (seed, sample_offset, noise_base)`` stands for
    the in-kernel noise (tests/_philox.py) when no noise buffer is given."""
    D = plan[0]
    if rng is not None and noise is None:
        from tests._philox import normals
        ints = etab.contiguous().view(torch.int32)
        n_idx = int(ints[:, 5].max()) + 1
        noise = torch.from_numpy(normals(rng[0], rng[1], x_in.shape[0], D, [rng[2] + i for i in range(n_idx)]))
    x = x_in.double()
    if in_shift is not None:
        x = x - in_shift.double()
    if in_scale is not None:
        x = x / in_scale.double()
    B = x.shape[0]
    ks = torch.zeros(7, B, D, dtype=torch.float64)
    kl = torch.zeros(7, B, dtype=torch.float64)
    x, lp = _run(plan, wpack, etab, etab.shape[0], x, cond, _tangents(plan, mode, probe, B), noise, ks, kl,
                 torch.zeros(B, dtype=torch.float64))
    if out_scale is not None:
        x = x * out_scale.double()
    if out_shift is not None:
        x = x + out_shift.double()
    return x, lp


def emulate_step(plan, wpack, etab, y, cond, probe, k1, kl1, lp0, mode, n_aux, first=0, count=0):
    """ff_mlp_ode_launch with the adaptive-step extras: slot 0 preloaded, n_aux linear combinations
    of the slots described by the two trailing rows.  Returns (aux [n_aux,B,D], aux_lp [n_aux,B]) fp32."""
    D = plan[0]
    x = y.double()
    B = x.shape[0]
    ks = torch.zeros(7, B, D, dtype=torch.float64)
    kl = torch.zeros(7, B, dtype=torch.float64)
    if k1 is not None:
        ks[0] = k1.double()
    if kl1 is not None:
        kl[0] = kl1.double()
    n = etab.shape[0] - 2
    x, _ = _run(plan, wpack, etab, n, x, cond, _tangents(plan, mode, probe, B, first, count), None, ks, kl,
                torch.zeros(B, dtype=torch.float64))
    e64 = etab.double()
    use_y = int(etab.contiguous().view(torch.int32)[n, 3])
    coefs = [e64[n, 8:15], e64[n, 16:23], e64[n + 1, 8:15], e64[n + 1, 16:23]]
    aux = torch.zeros(n_aux, B, D, dtype=torch.float64)
    aux_lp = torch.zeros(n_aux, B if mode else 0, dtype=torch.float64)
    l0 = lp0.double() if lp0 is not None else torch.zeros(B, dtype=torch.float64)
    for j in range(n_aux):
        uy = float((use_y >> j) & 1)
        aux[j] = uy * x + torch.einsum("s,sbd->bd", coefs[j], ks)
        if mode:
            aux_lp[j] = uy * l0 + torch.einsum("s,sb->b", coefs[j], kl)
    return aux.float(), aux_lp.float()


def emulate_rhs_jac(plan, wpack, rows, y, cond, first, count, jac):
    """flowfusion_amd::mlp_rhs_jac: one evaluation row + two auxiliary rows; returns rhs (fp32) and fills
    rows [first, first + count) of jac[b, j, i] = d rhs_i / d y_j."""
    D = plan[0]
    x = y.double()
    B = x.shape[0]
    ks = torch.zeros(7, B, D, dtype=torch.float64)
    kl = torch.zeros(7, B, dtype=torch.float64)
    _run(plan, wpack, rows, 1, x, cond, _tangents(plan, 2, None, B, first, count), None, ks, kl,
         torch.zeros(B, dtype=torch.float64), jac=jac, jac_first=first)
    return ks[0].float()
