"""CPU emulation of the mlp_ode kernel's *semantics* (TEST INFRASTRUCTURE).

Reads exactly what the kernel reads -- the packed weight buffer produced by ff_mlp_wpack and the
evaluation table produced by flowfusion_amd.solvers -- and performs the same stage bookkeeping in
float64 torch ops.  It lets the CPU-only test tier check all host-side logic (packing order, table
words, time reversal signs, stage slots, noise rows, tangent columns) against the oracle without a
GPU.  The lane-level MFMA mapping itself is only exercised by the `-m gpu` tests.
"""
import torch


def feat_of_reg(r, h):
    return 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * h


def decode_wpack(plan, wpack):
    """Invert the packing: dense (zero padded) matrices W1[H, 2*dregs + 2*cregs], hidden [H,H]+bias, out."""
    D, C, NH, H, dregs, cregs = plan[0], plan[1], plan[2], plan[3], plan[4], plan[5]
    NB = H // 32
    nob_out = (dregs + 15) // 16
    w = wpack.double()
    pos = 0

    def take(n):
        nonlocal pos
        out = w[pos:pos + n]
        pos += n
        return out

    def decode(KR, NOB, kdim):
        blk = take((KR // 4) * NOB * 256).view(KR // 4, NOB, 64, 4)
        M = torch.zeros(NOB * 32, kdim, dtype=torch.float64)
        for g in range(KR // 4):
            for q in range(4):
                r = 4 * g + q
                for h in (0, 1):
                    k = kmap(r, h)
                    if k is None:
                        continue
                    lanes = torch.arange(32) + 32 * h
                    for ob in range(NOB):
                        M[ob * 32:(ob + 1) * 32, k] = blk[g, ob, lanes, q]
        return M

    # first layer: columns [x dims (2*dregs) | cond dims (2*cregs)]
    def kmap(r, h):
        return feat_of_reg(r, h) if r < dregs else 2 * dregs + feat_of_reg(r - dregs, h)
    W1 = decode(dregs + cregs, NB, 2 * dregs + 2 * cregs)

    def kmap(r, h):   # noqa: F811
        return feat_of_reg(r, h)
    hidden = []
    for _ in range(NH - 1):
        Wl = decode(NB * 16, NB, H)
        bl = take(H)
        hidden.append((Wl, bl))
    Wo = decode(NB * 16, nob_out, H)
    bo = take(nob_out * 32)
    assert pos == w.numel(), (pos, w.numel())
    return W1, hidden, Wo, bo


def _silu(a):
    s = torch.sigmoid(a)
    return a * s, s + a * s * (1 - s)


def emulate(plan, wpack, etab, x_in, cond=None, probe=None, noise=None, mode=0,
            in_shift=None, in_scale=None, out_scale=None, out_shift=None):
    """Returns (x_out [B,D], dlogp [B]) in float64."""
    D, C, NH, H, dregs, cregs = plan[0], plan[1], plan[2], plan[3], plan[4], plan[5]
    W1, hidden, Wo, bo = decode_wpack(plan, wpack)
    W1x, W1c = W1[:, :D], W1[:, 2 * dregs:2 * dregs + C]
    etab64 = etab.double()
    ints = etab.contiguous().view(torch.int32)
    x = x_in.double()
    if in_shift is not None:
        x = x - in_shift.double()
    if in_scale is not None:
        x = x / in_scale.double()
    B = x.shape[0]
    if mode == 0:
        V = None
    elif mode == 1:
        V = probe.double()[:, None, :]                          # [B, T=1, D]
    else:
        V = torch.eye(D, dtype=torch.float64)[None].expand(B, D, D)
    ks = torch.zeros(6, B, D, dtype=torch.float64)
    kl = torch.zeros(6, B, dtype=torch.float64)
    lp = torch.zeros(B, dtype=torch.float64)
    cc = cond.double() @ W1c.T if C else 0.0
    for e in range(etab.shape[0]):
        a, b, gn = etab64[e, 0], etab64[e, 1], etab64[e, 2]
        flags, slot, nidx = int(ints[e, 3]), int(ints[e, 4]), int(ints[e, 5])
        cin, cout = etab64[e, 8:14], etab64[e, 16:22]
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
        rhs = a * y + b * net
        ks[slot] = rhs
        if V is not None:
            dnet = (dh @ Wo.T)[:, :, :D]                        # [B,T,D]
            div = a * (V * V).sum((1, 2)) + b * (V * dnet).sum((1, 2))
            kl[slot] = div
        if flags & 1:
            x = x + torch.einsum("s,sbd->bd", cout, ks)
            lp = lp + torch.einsum("s,sb->b", cout, kl)
        if flags & 2:
            x = x + gn * noise[nidx].double()
    if out_scale is not None:
        x = x * out_scale.double()
    if out_shift is not None:
        x = x + out_shift.double()
    return x, lp
