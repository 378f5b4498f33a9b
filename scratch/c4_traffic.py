"""Round 4: where do the 512-wide kernel's 1.4 TB of memory-side reads per config-4 launch come from?
One fused launch of a 64-d flow with N hidden layers of 512 (argv[1]: 5 = BASELINE configs[3], 4.47 MB of packed weights;
4 = 3.42 MB; 3 = 2.37 MB), 2^17 rows, 20-step RK4 (80 evaluations), to be run under `rocprofv3 --pmc FETCH_SIZE`
(scratch/c4_traffic.sh sums the counter).  Prediction if the traffic is an L2 CAPACITY miss of a stream that the 128
wavefronts of an XCD already share (not a per-wavefront re-read that LDS staging could remove):
    bytes = weights x evaluations x (tiles / 1024 in flight) rounds x 8 XCDs   while the set exceeds the 4 MiB L2,
    and close to nothing once it fits."""
import sys, torch
sys.path.insert(0, '/root/repo')
from flowfusion_amd.flow import ODEFlow
nh = int(sys.argv[1])
dev = 'cuda'
torch.manual_seed(0)
f = ODEFlow(64, [512] * nh).to(dev).eval()
net = f._net()
B = 1 << 17
tab = f._table(torch.tensor([1.0, 0.0]), "rk4", {"step_size": 1.0 / 20}, 0).to(dev)
x = torch.randn(B, 64, device=dev)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); net.integrate(x, tab, 0); e.record(); torch.cuda.synchronize()
mac = 65 * 512 + (nh - 1) * 512 * 512 + 512 * 64
wbytes = 4 * int(net.wpack(dev, 0).numel())
rounds = B / 16 / 1024
print(f"hidden layers {nh}: packed weights {wbytes / 1e6:.2f} MB, {tab.shape[0]} evaluations, {rounds:.0f} rounds of 1024 tiles, "
      f"launch {s.elapsed_time(e):.1f} ms; predicted if every XCD re-fetches the set per evaluation and round: "
      f"{wbytes * tab.shape[0] * rounds * 8 / 1e9:.2f} GB", flush=True)
