"""Accuracy of the three-way bf16 split (six products, fp32 accumulation on the bf16 MFMA path) for a 256x256
layer, against float64, next to plain fp32 and to fewer products.  Planning data for DESIGN.md section 8 item 1."""
import torch
torch.manual_seed(0)
dev = "cuda"
W = (torch.randn(256, 256, device=dev) / 16)
X = torch.randn(256, 4096, device=dev) * torch.rand(256, 1, device=dev) * 3
ref = W.double() @ X.double()
scale = (W.double().abs() @ X.double().abs())            # sum |w||x| : the natural error scale

def split3(t):
    mask = torch.tensor(-65536, dtype=torch.int32, device=t.device)       # 0xFFFF0000
    top = lambda v: (v.view(torch.int32) & mask).view(torch.float32)
    hi = top(t); r1 = t - hi; mid = top(r1); r2 = r1 - mid; lo = top(r2)
    return hi.bfloat16(), mid.bfloat16(), lo.bfloat16()

def mm(a, b):
    try:
        return torch.mm(a, b, out_dtype=torch.float32)
    except TypeError:
        return (a.float() @ b.float())            # fallback: fp32 accumulate of exact bf16 products (not the MFMA path)

Wh, Wm, Wl = split3(W)
Xh, Xm, Xl = split3(X)
print("split exact:", float((Wh.float() + Wm.float() + Wl.float() - W).abs().max()), float((Xh.float() + Xm.float() + Xl.float() - X).abs().max()))
def err(y):
    return float(((y.double() - ref).abs() / scale).max()), float(((y.double() - ref).abs() / scale).mean())
print("fp32 matmul              max/mean rel-to-sum|w||x|: %.2e %.2e" % err(W @ X))
six = mm(Wh, Xh) + (mm(Wh, Xm) + mm(Wm, Xh)) + (mm(Wh, Xl) + mm(Wm, Xm) + mm(Wl, Xh))
print("6 products (bf16 MFMA)                              : %.2e %.2e" % err(six))
three = mm(Wh, Xh) + (mm(Wh, Xm) + mm(Wm, Xh))
print("3 products                                          : %.2e %.2e" % err(three))
print("1 product (plain bf16)                              : %.2e %.2e" % err(mm(Wh, Xh)))
nine = six + (mm(Wm, Xl) + mm(Wl, Xm)) + mm(Wl, Xl)
print("9 products                                          : %.2e %.2e" % err(nine))
