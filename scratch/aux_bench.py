"""HBM rates of the streaming helpers (csrc/ff_aux.hip) at solver-sized arrays: algorithmic bytes / HIP-event time,
against the 8 TB/s roofline and against torch's own copy kernel on the same box (what a streaming kernel reaches
here in practice).  python scratch/aux_bench.py [log2 of the batch, default 22] -> one JSON line."""
import json
import sys

import torch

sys.path.insert(0, ".")
from flowfusion_amd import _native  # noqa: E402

DEV = torch.device("cuda", 0)
PEAK = 8000.0


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    lb = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    B, D = 1 << lb, 16
    n = B * D
    x = torch.randn(B, D, device=DEV)
    ks = [torch.randn(B, D, device=DEV) for _ in range(7)]
    out = torch.empty_like(x)
    rows = []

    def rec(name, nbytes, ms):
        gbs = nbytes / (ms * 1e-3) / 1e9
        rows.append({"kernel": name, "bytes": nbytes, "ms": round(ms, 4), "GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 3)})

    ms = timed(lambda: out.copy_(x))
    rec("torch copy_ (yardstick: 4 B read + 4 B written per element)", 8 * n, ms)
    for terms in (1, 4, 7):
        coefs = [0.1 * (i + 1) for i in range(terms)]
        ms = timed(lambda: _native.stage_combine(out, x, ks[:terms], coefs, 1.0))
        rec(f"ff_stage_combine, {terms} terms + x", 4 * n * (terms + 2), ms)
    ms = timed(lambda: _native.normal_fill(B, D, 1234, 0, DEV))
    rec("ff_normal_fill (4 B written per element; Philox4x32-10 + Box-Muller per 4 elements)", 4 * n, ms)
    # the adaptive step's norms: err / max(|y0|, |y1|), and the finiteness check of y1 (5 arrays read)
    y0, y1, err = x, ks[0], ks[1]
    ms = timed(lambda: _native.scaled_rms([(err, None, y0, y1)], 1e-5, 1e-5, check=y1))
    rec("ff_scaled_rms, 1 term + finiteness check of its second scale array (3 array reads per element, one pass)", 4 * n * 3, ms)
    ms = timed(lambda: _native.scaled_rms([(err, None, y0, y1), (ks[2], ks[3], y0, None), (ks[4], None, y0, None)], 1e-5, 1e-5, check=y1))
    rec("ff_scaled_rms, 3 terms + check folded into the first (9 array reads per element)", 4 * n * 9, ms)
    print(json.dumps({"batch": B, "dim": D, "rows": rows}))
    for r in rows:
        print(f"{r['kernel'][:80]:80s} {r['ms']:8.3f} ms  {r['GBps']:8.1f} GB/s  {r['frac_of_8TBps']:.3f}", file=sys.stderr)


if __name__ == "__main__":
    main()
