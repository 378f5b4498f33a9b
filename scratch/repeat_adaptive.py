"""Repeat one default-argument adaptive log_prob and print a fingerprint per repeat (attempt counts, checksum)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowfusion_amd import device_adaptive
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel
rank = int(os.environ.get("RANK", "0"))
dev = torch.device("cuda", 0)
torch.manual_seed(2)
hm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True, hutchinson=True).eval().to(dev)
x = torch.randn(int(os.environ.get("ROWS", "3001")), 16, device=dev) * 0.8
if os.environ.get("POISON"):
    device_adaptive.POISON = float(os.environ["POISON"])
seen = {}
for i in range(int(os.environ.get("REPS", "40"))):
    if os.environ.get("TRACE"):
        device_adaptive.TRACE = []
    r = hm.log_prob(x, probe="philox", seed=9)
    key = (hm.last_solver_stats["attempts"], hm.last_solver_stats["accepted"], float(r.double().sum()))
    seen.setdefault(key, []).append(i)
    if os.environ.get("TRACE") and i < 3:
        print(rank, i, [(round(t[3], 9), round(t[4], 7)) for t in device_adaptive.TRACE[:4]], flush=True)
for k, v in sorted(seen.items(), key=lambda kv: -len(kv[1]))[:4]:
    print(f"rank {rank}: {len(seen)} distinct; {k} at repeats {v[:12]}{'...' if len(v) > 12 else ''} ({len(v)})", flush=True)
