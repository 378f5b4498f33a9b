"""Child of tests/test_gpu_device_adaptive.py::test_global_step_control_over_ranks: one rank of a gloo group whose ranks all
use cuda:0 (RCCL refuses two ranks on one device; on a node the same code runs over RCCL).  Every rank solves its shard
under distributed.global_step_control and the whole batch on its own, and writes one JSON record under $FF_RESULT_DIR."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flowfusion_amd import flow as Fm
    from flowfusion_amd.diffusion import MLP, VESDE, VPSDE, ScoreModel
    from flowfusion_amd.distributed import global_step_control, shard_bounds
    dev = torch.device("cuda", 0)
    out = {"rank": rank}

    def err(a, b):
        return float((a - b).abs().max() / b.abs().max().clamp_min(1.0))

    def case(name, solve, n):
        """solve(lo, hi) -> (tensor, stats).  Whole batch alone, own shard alone, own shard under the global control."""
        lo, hi = shard_bounds(n, world, rank)
        whole, s_whole = solve(0, n)
        alone, s_alone = solve(lo, hi)
        with global_step_control():
            mine, s_mine = solve(lo, hi)
        out[name] = {"whole": s_whole, "alone": s_alone, "global": s_mine, "err_global": err(mine, whole[lo:hi]),
                     "err_alone": err(alone, whole[lo:hi])}

    torch.manual_seed(0)
    nb = ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).eval().to(dev)
    n = 3001
    z = torch.randn(n, 2, device=dev) * 3.0
    z[: n // 3] *= 4.0                                   # the first shard is the hard one: the shards' own norms differ
    pts = torch.randn(n, 2, device=dev) * 0.5
    pts[: n // 3] *= 3.0

    def sample(lo, hi):
        y, _ = nb.sample_ode_from_base(z[lo:hi].contiguous())
        return y, dict(nb.last_solver_stats)

    def logp(lo, hi):
        y = nb.log_prob(pts[lo:hi].contiguous())
        return y, dict(nb.last_solver_stats)

    case("sample_ve_2d", sample, n)
    case("log_prob_ve_2d_exact", logp, n)

    torch.manual_seed(1)
    f = Fm.ODEFlow(8, [128, 128]).to(dev).eval()
    xT = torch.randn(n, 8, device=dev)
    xT[: n // 3] *= 3.0

    def flow_sample(lo, hi):
        y = f.sample(xT[lo:hi].contiguous())
        return y, dict(f.last_solver_stats)

    case("flow_sample_8d", flow_sample, n)

    # the host step controller takes the same exchange (what falls back to it: other modules, dopri8, ...)
    os.environ["FF_HOST_CONTROLLER"] = "1"
    case("sample_ve_2d_host_controller", sample, n)
    os.environ.pop("FF_HOST_CONTROLLER")

    # the sharded forms of the public calls: base samples / probe keyed by the global row, whole-batch step control
    from flowfusion_amd import _native
    from flowfusion_amd.distributed import log_prob_sharded, sample_ode_sharded
    lo, hi = shard_bounds(n, world, rank)
    mine, span = sample_ode_sharded(nb, n, 2, seed=3, gather=False)
    whole, _ = nb.sample_ode_from_base(_native.normal_fill(n, 2, 3, 0, dev))
    out["sample_ode_sharded"] = {"span_ok": span == (lo, hi), "err": err(mine, whole[lo:hi])}
    torch.manual_seed(2)
    hm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True, hutchinson=True).eval().to(dev)
    x16 = torch.randn(n, 16, device=dev) * 0.8
    mine, span = log_prob_sharded(hm, x16, seed=9, gather=False)                 # default arguments: adaptive dopri5
    s_mine = dict(hm.last_solver_stats)
    whole = hm.log_prob(x16, probe="philox", seed=9)
    out["log_prob_sharded_hutch"] = {"span_ok": span == (lo, hi), "err": err(mine, whole[lo:hi]),
                                     "same_steps": s_mine["attempts"] == hm.last_solver_stats["attempts"],
                                     "global": s_mine, "whole": dict(hm.last_solver_stats)}

    # the flows' sharded entry points (BASELINE configs[3] is worded "sharded over 8xMI355X"): default arguments = adaptive
    # dopri5 under the whole-batch step control; Hutchinson probe keyed by the global row
    from flowfusion_amd.distributed import flow_log_prob_sharded, flow_sample_sharded
    mine, span = flow_sample_sharded(f, n, seed=5, gather=False)
    s_mine = dict(f.last_solver_stats)
    whole = f.sample(_native.normal_fill(n, 8, 5, 0, dev))
    out["flow_sample_sharded"] = {"span_ok": span == (lo, hi), "err": err(mine, whole[lo:hi]),
                                  "same_steps": s_mine["attempts"] == f.last_solver_stats["attempts"]}
    xf = torch.randn(n, 8, device=dev) * 0.7
    mine, span = flow_log_prob_sharded(f, xf, seed=4, gather=False, hutchinson=True, atol=1e-4, rtol=1e-4)
    s_mine = dict(f.last_solver_stats)
    whole = f.log_prob(xf, atol=1e-4, rtol=1e-4, hutchinson=True, probe="philox", seed=4)
    out["flow_log_prob_sharded_hutch"] = {"span_ok": span == (lo, hi), "err": err(mine, whole[lo:hi]),
                                          "same_steps": s_mine["attempts"] == f.last_solver_stats["attempts"]}
    gathered = flow_sample_sharded(f, n, seed=5, method="rk4", options={"step_size": 0.125})     # fixed grid + the one all-gather
    out["flow_sample_sharded_gathered"] = bool(torch.equal(
        gathered, f.sample(_native.normal_fill(n, 8, 5, 0, dev), method="rk4", options={"step_size": 0.125})))
    # fewer rows than ranks under the whole-batch control: every rank raises before anyone enters a collective
    try:
        flow_sample_sharded(f, world - 1, seed=5)
        out["too_few_rows_raises"] = False
    except ValueError:
        out["too_few_rows_raises"] = True

    # an empty shard cannot take part
    try:
        with global_step_control():
            nb.sample_ode_from_base(z[:0].contiguous())
        out["empty_raises"] = False
    except ValueError:
        out["empty_raises"] = True
    dist.barrier()
    with open(os.path.join(os.environ["FF_RESULT_DIR"], f"rank{rank}.json"), "w") as fh:      # (stdout lines of the ranks interleave)
        json.dump(out, fh)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
