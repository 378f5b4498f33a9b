"""Child of tests/test_gpu_full_configs.py::test_rccl_single_rank: a ONE-rank "nccl" (= RCCL) process group on cuda:0.
No second GPU exists on the test box and RCCL refuses two ranks on one device, so this is as far as RCCL itself can be
exercised here: communicator set-up with device_id, every collective bench.py / distributed.py issue (all_reduce of
float64 device tensors, all_gather_into_tensor, all_gather, barrier), and the adaptive path's exchange hook enqueueing its
all-reduce from inside the C driver's callback."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {}
    t = torch.full((8,), 1.5, dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    out["all_reduce_f64"] = bool((t == 1.5).all())
    x = torch.randn(1000, 16, device=dev)
    g = torch.empty_like(x)
    dist.all_gather_into_tensor(g, x)
    out["all_gather_into_tensor"] = bool(torch.equal(g, x))
    lst = [torch.zeros(2, dtype=torch.float64, device=dev)]
    dist.all_gather(lst, torch.tensor([1.0, 2.0], dtype=torch.float64, device=dev))
    out["all_gather"] = lst[0].tolist() == [1.0, 2.0]
    dist.barrier()
    from flowfusion_amd import _native
    from flowfusion_amd.diffusion import MLP, VESDE, VPSDE, ScoreModel
    from flowfusion_amd.distributed import (gather_rows, global_step_control, log_prob_sharded, sample_ode_sharded,
                                            sample_sde_sharded, sum_over_ranks, sum_over_ranks_)
    out["sum_over_ranks"] = sum_over_ranks([1.0, 2.5], dev) == [1.0, 2.5]
    out["sum_over_ranks_"] = bool((sum_over_ranks_(torch.full((8,), 2.0, dtype=torch.float64, device=dev)) == 2.0).all())
    out["gather_rows"] = bool(torch.equal(gather_rows(x, 1000), x))
    torch.manual_seed(0)
    nb = ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).eval().to(dev)
    z = torch.randn(5000, 2, device=dev) * 3
    ref, _ = nb.sample_ode_from_base(z)
    st_ref = dict(nb.last_solver_stats)
    with global_step_control():                        # the hook's all-reduce runs under RCCL, enqueued between two launches
        got, _ = nb.sample_ode_from_base(z)
    out["global_control_device"] = bool(torch.equal(got, ref)) and dict(nb.last_solver_stats) == st_ref
    os.environ["FF_HOST_CONTROLLER"] = "1"
    ref_h, _ = nb.sample_ode_from_base(z)
    with global_step_control():
        got_h, _ = nb.sample_ode_from_base(z)
    os.environ.pop("FF_HOST_CONTROLLER")
    out["global_control_host"] = float((got_h - ref_h).abs().max() / ref_h.abs().max()) < 1e-5
    whole = sample_ode_sharded(nb, 4000, 2, seed=3)
    out["sample_ode_sharded"] = bool(torch.equal(whole, nb.sample_ode_from_base(_native.normal_fill(4000, 2, 3, 0, dev))[0]))
    torch.manual_seed(2)
    hm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True, hutchinson=True).eval().to(dev)
    x16 = torch.randn(3001, 16, device=dev) * 0.8
    out["log_prob_sharded"] = bool(torch.equal(log_prob_sharded(hm, x16, seed=9), hm.log_prob(x16, probe="philox", seed=9)))
    s = sample_sde_sharded(hm, (2000, 16), steps=10, seed=1)
    out["sample_sde_sharded"] = bool(torch.isfinite(s).all()) and tuple(s.shape) == (2000, 16)
    from flowfusion_amd import flow as Fm
    from flowfusion_amd.distributed import flow_log_prob_sharded, flow_sample_sharded
    torch.manual_seed(4)
    fl = Fm.ConditionalODEFlow(6, 2, [128, 128]).eval().to(dev)
    cf = torch.randn(2500, 2, device=dev)
    got = flow_sample_sharded(fl, 2500, seed=5, conditional=cf)
    out["flow_sample_sharded"] = bool(torch.equal(got, fl.sample(_native.normal_fill(2500, 6, 5, 0, dev), cf)))
    xf = torch.randn(2500, 6, device=dev) * 0.7
    got = flow_log_prob_sharded(fl, xf, cf, seed=8, hutchinson=True)
    out["flow_log_prob_sharded"] = bool(torch.equal(got, fl.log_prob(xf, cf, hutchinson=True, probe="philox", seed=8)))
    dist.barrier()
    dist.destroy_process_group()
    with open(os.path.join(os.environ["FF_RESULT_DIR"], "rccl.json"), "w") as fh:
        json.dump(out, fh)


if __name__ == "__main__":
    main()
