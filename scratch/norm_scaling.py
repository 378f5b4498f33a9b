"""ff_scaled_rms (1 term + folded check) and torch.copy_ over array sizes: fixed cost vs per-byte rate."""
import sys, torch
sys.path.insert(0, ".")
from flowfusion_amd import _native
dev = torch.device("cuda", 0)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for lb in (16, 18, 20, 22, 23, 24):
    n = (1 << lb) * 16
    e, y0, y1, out = (torch.randn(n, device=dev) for _ in range(4))
    t_n = timed(lambda: _native.scaled_rms([(e, None, y0, y1)], 1e-5, 1e-5, check=y1))
    t_c = timed(lambda: out.copy_(e))
    t_s = timed(lambda: torch.sum(e))
    print(f"2^{lb} x16: norm {t_n*1e3:8.1f} us ({12*n/t_n/1e9:7.1f} GB/s)   copy {t_c*1e3:8.1f} us ({8*n/t_c/1e9:7.1f} GB/s)   torch.sum {t_s*1e3:8.1f} us ({4*n/t_s/1e9:7.1f} GB/s)", flush=True)
