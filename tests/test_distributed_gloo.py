"""world_size-2 run of the sharding / gather logic on the CPU (gloo).  The compute function is a
stand-in row-wise map: what is under test is the partition and the single collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flowfusion_amd.distributed import gather_rows, log_prob_sharded, run_sharded, shard_bounds, shard_sizes


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 8, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = shard_sizes(n, world)
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        x = torch.randn(n, 5)
        c = torch.randn(n, 2)
        fn = lambda a, b: torch.tanh(a) * 2 + b.sum(1, keepdim=True)
        full = run_sharded(fn, [x, c])
        ok = torch.equal(full, fn(x, c))
        local, (lo, hi) = run_sharded(fn, [x, c], gather=False)
        ok &= torch.equal(local, fn(x, c)[lo:hi]) and (lo, hi) == shard_bounds(n, world, rank)
        pair = run_sharded(lambda a, b: (a + 1, b * 2), [x, c])
        ok &= torch.equal(pair[0], x + 1) and torch.equal(pair[1], c * 2)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 64), (2, 37), (8, 64), (8, 37), (8, 5)])     # even, ragged, fewer rows than ranks
def test_sharding_and_gather_over_gloo(world, n):
    """world 2 and world 8 (the node the north star names): bench.py's N > 1 branch and distributed.py reduce to
    `shard_bounds` + one all-gather, whatever the world size."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    results = dict(q.get(timeout=5) for _ in range(world))
    assert results == {r: True for r in range(world)}


class _RowKeyedModel:
    """Stand-in for a Hutchinson ScoreModel: log_prob(rows) depends on each row's values, its conditional and the GLOBAL
    row index it is told (sample_offset + r) -- what the counter-based probe makes true of the real one."""
    hutch = True

    def log_prob(self, rows, conditional=None, probe="torch", seed=None, sample_offset=0, **solver):
        assert probe == "philox" and solver == {"method": "rk4", "options": {"step_size": 0.1}}
        g = torch.arange(rows.shape[0], dtype=torch.float32) + float(sample_offset)
        c = 0.0 if conditional is None else conditional.sum(1)
        return (rows.sum(1) + 1000.0 * g + float(seed) + c).view(-1, 1)


def _logp_worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        x, c = torch.randn(n, 5), torch.randn(n, 2)
        m = _RowKeyedModel()
        kw = {"method": "rk4", "options": {"step_size": 0.1}}
        expect = m.log_prob(x, conditional=c, probe="philox", seed=9, sample_offset=0, **kw)
        full = log_prob_sharded(m, x, c, seed=9, **kw)
        ok = torch.equal(full, expect)
        lo, hi = shard_bounds(n, world, rank)
        mine = log_prob_sharded(m, local_x=x[lo:hi], local_conditional=c[lo:hi], n_total=n, seed=9, **kw)
        ok &= torch.equal(mine, expect)
        local, span = log_prob_sharded(m, x, c, seed=9, gather=False, **kw)
        ok &= span == (lo, hi) and torch.equal(local, expect[lo:hi])
        try:
            log_prob_sharded(m, local_x=x[: hi - lo + 1], n_total=n, seed=9, **kw)
            ok = False
        except ValueError:
            pass
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 37), (8, 64), (8, 5)])
def test_log_prob_sharded_over_gloo(world, n):
    """The sharded log-density helper: every rank is told the global index of its first row (the key of the probe
    stream), full-batch and local-rows calling conventions agree, one all-gather."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_logp_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    results = dict(q.get(timeout=5) for _ in range(world))
    assert results == {r: True for r in range(world)}


def _global_control_worker(rank, world, port, q):
    """An adaptive dopri5 solve (host controller on the CPU kernel-semantics emulator) of a batch cut over the ranks: under
    distributed.global_step_control every rank walks the steps of the whole-batch solve."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flowfusion_amd import _native, adaptive
        from flowfusion_amd import diffusion as D
        from flowfusion_amd.distributed import global_step_control
        from flowfusion_amd.fused import MODE_HUTCH
        from tests import _emulator as E
        torch.manual_seed(3)
        sm = D.ScoreModel(D.MLP(3, 0, 8, [64, 64]), D.VESDE(), no_sigma=False).eval()
        net = sm._net()
        n = 11
        x = torch.randn(n, 3)
        x[:4] *= 6.0                                        # the first shard carries the largest errors
        e = torch.sign(torch.randn(n, 3))
        eps = float(torch.tensor(float(sm.sde.epsilon), dtype=torch.float32))
        sched = lambda tr: sm._schedule(tr, "ode")[:3]
        plan = _native.plan_words(net.plan(MODE_HUTCH))
        wpack = net.wpack("cpu", MODE_HUTCH)

        def solve(lo, hi):
            pr = e[lo:hi]
            launcher = lambda y, k1, kl1, lp0, etab, n_aux, first, count: E.emulate_step(
                plan, wpack, etab, y, None, pr, k1, kl1, lp0, MODE_HUTCH, n_aux, first, count)
            step = net.make_step(sched, 1.0, MODE_HUTCH, "cpu", probe=pr, launcher=launcher)
            solver = adaptive.Dopri5(step, True, 1e-5, 1e-5, {"min_step": 1e-9})
            y, lp = solver.integrate(eps, 1.0, x[lo:hi], torch.zeros(hi - lo))
            return y, lp, (solver.n_attempts, solver.n_accepted)

        lo, hi = shard_bounds(n, world, rank)
        yw, lw, sw = solve(0, n)
        ya, la, sa = solve(lo, hi)
        with global_step_control():
            yg, lg, sg = solve(lo, hi)
        err = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1.0))
        q.put((rank, {"whole": sw, "alone": sa, "global": sg,
                      "err_global": max(err(yg, yw[lo:hi]), err(lg, lw[lo:hi])),
                      "err_alone": max(err(ya, yw[lo:hi]), err(la, lw[lo:hi]))}))
    finally:
        dist.destroy_process_group()


def test_global_step_control_over_gloo(built_library):
    """The one exchange step of the path (an adaptive solve's batch-global error norm) on the CPU: two ranks, the host step
    controller on the emulated kernels.  Under the context both ranks attempt / accept the whole-batch solve's steps and
    reproduce its rows to the rounding of the norms (fp32 mean of squares on one rank vs float64 sums that met: the step
    sizes differ in their last bits)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_global_control_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    results = dict(q.get(timeout=5) for _ in range(world))
    assert set(results) == {0, 1}
    for r, c in results.items():
        assert tuple(c["global"]) == tuple(c["whole"]), (r, c)
        assert c["err_global"] < 2e-5, (r, c)
        assert c["err_alone"] < 5e-3, (r, c)
    assert any(tuple(c["alone"]) != tuple(c["whole"]) or c["err_alone"] > 100 * max(c["err_global"], 1e-12) for c in results.values())
