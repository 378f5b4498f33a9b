"""Child of tests/test_gpu_device_adaptive.py::test_cooperative_twin_is_deterministic_when_the_card_is_shared: repeats one
default-argument adaptive log_prob at a cooperative-twin batch while its sibling processes do the same on the same card, and
writes the distinct (attempts, accepted, checksum) fingerprints it saw under $FF_RESULT_DIR."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel
    rank = int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", 0)
    torch.manual_seed(2)
    hm = ScoreModel(MLP(16, 0, 8, [256] * 4), VPSDE(), no_sigma=True, hutchinson=True).eval().to(dev)
    x = torch.randn(3001, 16, device=dev) * 0.8
    seen = {}
    for ctrl in ("device", "host"):
        if ctrl == "host":
            os.environ["FF_HOST_CONTROLLER"] = "1"
        for _ in range(25):
            r = hm.log_prob(x, probe="philox", seed=9)
            key = f"{ctrl} {hm.last_solver_stats['attempts']} {hm.last_solver_stats['accepted']} {float(r.double().sum())!r}"
            seen[key] = seen.get(key, 0) + 1
    os.environ.pop("FF_HOST_CONTROLLER", None)
    # an odd number of hidden layers (the notebook's 3 x 128): the exchange buffers of consecutive evaluations
    from flowfusion_amd.diffusion import VESDE
    torch.manual_seed(0)
    nb = ScoreModel(MLP(2, 0, 8, [128] * 3), VESDE()).eval().to(dev)
    z = torch.randn(1000, 2, device=dev) * 3
    for _ in range(25):
        r, _ = nb.sample_ode_from_base(z)
        key = f"odd {nb.last_solver_stats['attempts']} {nb.last_solver_stats['accepted']} {float(r.double().sum())!r}"
        seen[key] = seen.get(key, 0) + 1
    with open(os.path.join(os.environ["FF_RESULT_DIR"], f"contention{rank}.json"), "w") as fh:
        json.dump(seen, fh)


if __name__ == "__main__":
    main()
