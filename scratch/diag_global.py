"""Diagnostic: step sequences (t, dt, error ratio per attempt) of a whole-batch adaptive solve and of the same solve sharded
under distributed.global_step_control (run with torch.distributed.run, gloo, all ranks on cuda:0)."""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from flowfusion_amd import device_adaptive
from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel
from flowfusion_amd.distributed import global_step_control, shard_bounds
dev = torch.device("cuda", 0)
torch.manual_seed(2)
hm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True, hutchinson=True).eval().to(dev)
n = 3001
x16 = torch.randn(n, 16, device=dev) * 0.8
lo, hi = shard_bounds(n, world, rank)
if os.environ.get("POISON"):
    device_adaptive.POISON = float(os.environ["POISON"])
for r in range(world):            # one process on the card at a time
    if r == rank or not os.environ.get("SERIAL"):
        device_adaptive.TRACE = []
        whole = hm.log_prob(x16, probe="philox", seed=9)
        tw = device_adaptive.TRACE
        torch.cuda.synchronize()
    if os.environ.get("SERIAL"):
        dist.barrier()
    else:
        break
device_adaptive.TRACE = []
with global_step_control():
    mine = hm.log_prob(x16[lo:hi].contiguous(), probe="philox", seed=9, sample_offset=lo)
tg = device_adaptive.TRACE
device_adaptive.TRACE = None
if rank == 0:
    for a, b in zip(tw, tg):
        print("whole", a, "| global", b)
    print(len(tw), len(tg), float((mine - whole[lo:hi]).abs().max()))
dist.barrier()
dist.destroy_process_group()
