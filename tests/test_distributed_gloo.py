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


# ---- the flows' sharded entry points (BASELINE configs[3] is worded "sharded over 8xMI355X") ---------------------------
class _RowKeyedFlow:
    """Stand-in for ODEFlow / ConditionalODEFlow: results depend on each row's values, its conditional and -- for the
    Hutchinson log-density -- the GLOBAL row it is told; records whether the whole-batch step control was on."""
    target_dimension = 3

    def __init__(self):
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.controlled = []

    def parameters(self):
        return iter([self.w])

    def _note(self):
        from flowfusion_amd.distributed import step_control_group
        self.controlled.append(step_control_group()[0])

    def sample(self, xT, conditional=None, **solver):
        self._note()
        c = 0.0 if conditional is None else conditional.sum(1, keepdim=True)
        return torch.tanh(xT) * 2 + c + (1.0 if solver.get("method") == "rk4" else 0.0)

    def log_prob(self, x, conditional=None, hutchinson=False, probe="torch", seed=None, sample_offset=0, **solver):
        self._note()
        assert (probe == "philox") == bool(hutchinson)
        g = (torch.arange(x.shape[0], dtype=torch.float32) + float(sample_offset)) * (1000.0 if hutchinson else 0.0)
        c = 0.0 if conditional is None else conditional.sum(1)
        return x.sum(1) + g + (float(seed) if hutchinson else 0.0) + c


def _cpu_normal_fill(batch, dim, seed, sample_offset, device, noise_index=0xFFFFFFFF, scale=1.0):
    """ff_normal_fill's stream on the CPU (tests/_philox.py restates it): keyed by the global row, like the device's."""
    from tests import _philox
    return torch.from_numpy(_philox.normals(seed, sample_offset, batch, dim, [noise_index])[0].copy()) * scale


def _flow_worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flowfusion_amd import _native
        from flowfusion_amd.distributed import flow_log_prob_sharded, flow_sample_sharded
        _native.normal_fill = _cpu_normal_fill                       # (the product draws on the device; no GPU here)
        torch.manual_seed(0)
        x, c = torch.randn(n, 3), torch.randn(n, 2)
        lo, hi = shard_bounds(n, world, rank)
        f = _RowKeyedFlow()
        fixed = {"method": "rk4", "options": {"step_size": 0.1}}
        base = _cpu_normal_fill(n, 3, 5, 0, "cpu")
        ok = True
        # sample: every world size transports the same base points; full / local conditional; gather or keep the shard
        ok &= torch.equal(flow_sample_sharded(f, n, seed=5, **fixed), f.sample(base, **fixed))
        ok &= torch.equal(flow_sample_sharded(f, n, seed=5, conditional=c, **fixed), f.sample(base, c, **fixed))
        ok &= torch.equal(flow_sample_sharded(f, n, seed=5, local_conditional=c[lo:hi], **fixed), f.sample(base, c, **fixed))
        local, span = flow_sample_sharded(f, n, seed=5, conditional=c, gather=False, **fixed)
        ok &= span == (lo, hi) and torch.equal(local, f.sample(base, c, **fixed)[lo:hi])
        ok &= not any(f.controlled)                                  # fixed grids: no exchange, empty shards are fine
        # log_prob: exact trace (no random numbers) and Hutchinson (probe keyed by the global row)
        ok &= torch.equal(flow_log_prob_sharded(f, x, c, **fixed), f.log_prob(x, c, **fixed))
        want = f.log_prob(x, c, hutchinson=True, probe="philox", seed=9, sample_offset=0, **fixed)
        ok &= torch.equal(flow_log_prob_sharded(f, x, c, seed=9, hutchinson=True, **fixed), want)
        ok &= torch.equal(flow_log_prob_sharded(f, local_x=x[lo:hi], local_conditional=c[lo:hi], n_total=n, seed=9,
                                                hutchinson=True, **fixed), want)
        for bad in (dict(x=x, local_x=x[lo:hi]), dict(local_x=x[lo:hi]), dict(local_x=x[: hi - lo + 1], n_total=n),
                    dict(x=x, local_conditional=c[lo:hi]), dict(local_x=x[lo:hi], n_total=n, conditional=c)):
            try:
                flow_log_prob_sharded(f, **bad, **fixed)
                ok = False
            except ValueError:
                pass
        # the reference's default (adaptive dopri5): whole-batch step control when every rank has a row, a ValueError on
        # EVERY rank -- before anyone enters a collective -- when some rank has none; global_control=False needs neither
        f.controlled.clear()
        if n >= world:
            flow_sample_sharded(f, n, seed=5)
            flow_log_prob_sharded(f, x, c)
            ok &= f.controlled == [world > 1, world > 1]
        else:
            for call in (lambda: flow_sample_sharded(f, n, seed=5), lambda: flow_log_prob_sharded(f, x, c),
                         lambda: log_prob_sharded(_RowKeyedModel(), x, c, seed=9)):
                try:
                    call()
                    ok = False
                except ValueError as e:
                    ok &= "at least one row per rank" in str(e)
            ok &= f.controlled == []
        f.controlled.clear()
        got = flow_sample_sharded(f, n, seed=5, global_control=False)
        ok &= f.controlled == [False] and torch.equal(got, f.sample(base))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 64), (2, 37), (8, 64), (8, 37), (8, 5)])     # even, ragged, fewer rows than ranks
def test_flow_sharded_entry_points_over_gloo(world, n):
    """distributed.flow_sample_sharded / flow_log_prob_sharded: base samples and the Hutchinson probe keyed by the GLOBAL
    row, the raw conditional sliced per rank, one all-gather at the end, whole-batch step control for adaptive methods --
    and a clean ValueError on every rank when a shard would be empty under that control (an empty shard has no launch to
    hang the exchange on; its peers would wait in the all-reduce for ever)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flow_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    results = dict(q.get(timeout=5) for _ in range(world))
    assert results == {r: True for r in range(world)}


def test_step_control_context_is_per_thread():
    """`global_step_control` is a context variable: a solve on another host thread of the process does not inherit it."""
    import threading
    from flowfusion_amd import distributed as Dd
    seen = {}
    token = Dd._STEP_CONTROL.set((True, "g"))
    try:
        t = threading.Thread(target=lambda: seen.setdefault("other", Dd.step_control_group()))
        t.start()
        t.join()
        seen["here"] = Dd.step_control_group()
    finally:
        Dd._STEP_CONTROL.reset(token)
    assert seen == {"other": (False, None), "here": (True, "g")} and Dd.step_control_group() == (False, None)
