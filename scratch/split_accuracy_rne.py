"""Accuracy of bf16 splits of an fp32 layer product (256-term dot products), CPU / float64 emulation of the bf16 MFMA
(exact products, wide accumulation): truncation vs round-to-nearest parts, and which of the six products can go.
Relative to sum |w||x|, max and mean over 256 x 2048 outputs."""
import torch
torch.manual_seed(0)
W = (torch.randn(256, 256) / 16)
X = torch.randn(256, 2048) * torch.rand(256, 1) * 3
ref = W.double() @ X.double()
scale = W.double().abs() @ X.double().abs()

def top(v):
    return (v.view(torch.int32) & torch.tensor(-65536, dtype=torch.int32)).view(torch.float32)

def split_trunc(t):
    hi = top(t); r1 = t - hi; mid = top(r1); r2 = r1 - mid; lo = top(r2)
    return hi, mid, lo

def split_rne(t):
    hi = t.bfloat16().float(); r1 = t - hi; mid = r1.bfloat16().float(); r2 = r1 - mid; lo = r2.bfloat16().float()
    return hi, mid, lo

def err(y):
    e = (y - ref).abs() / scale
    return "%.2e %.2e" % (float(e.max()), float(e.mean()))

mm = lambda a, b: a.double() @ b.double()
print("fp32 matmul                         :", err((W @ X).double()))
for name, sp in (("truncation", split_trunc), ("round-to-nearest", split_rne)):
    Wh, Wm, Wl = sp(W); Xh, Xm, Xl = sp(X)
    hh, hm, mh, hl, mmm, lh = mm(Wh, Xh), mm(Wh, Xm), mm(Wm, Xh), mm(Wh, Xl), mm(Wm, Xm), mm(Wl, Xh)
    print(f"{name:17s} 6 products         :", err(hh + hm + mh + hl + mmm + lh))
    print(f"{name:17s} 5 (no mid.mid)     :", err(hh + hm + mh + hl + lh))
    print(f"{name:17s} 4 (no mm, no wl.xh):", err(hh + hm + mh + hl))
    print(f"{name:17s} 4 (no mm, no wh.xl):", err(hh + hm + mh + lh))
    print(f"{name:17s} 3 (hh hm mh)       :", err(hh + hm + mh))
    print(f"{name:17s} 2 (hh hm): x 16 bit, w 8 bit:", err(hh + hm))
