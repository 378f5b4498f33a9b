import time, torch
dev = "cuda"
for N, r in ((10000, 2), (100000, 2), (100000, 4), (1000000, 2)):
    Y = torch.randn(N, 16, r, device=dev)
    torch.linalg.qr(Y[:100], mode="reduced"); torch.cuda.synchronize()
    t0 = time.perf_counter(); Q, R = torch.linalg.qr(Y, mode="reduced"); torch.cuda.synchronize()
    print(f"linalg.qr N={N} r={r}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    if N >= 100000 and r == 4:
        E = torch.eye(r, device=dev).expand(N, r, r).contiguous()
        t0 = time.perf_counter(); torch.linalg.solve_triangular(R, E, upper=True); torch.cuda.synchronize()
        print(f"  solve_triangular: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
