"""The N > 1 path on CPU: gloo ranks shard (level, molecule) units and collect them.  The
per-rank compute is the CPU oracle here (test infrastructure); on GPUs it is the HIP engine
(pylbl_amd.distributed.ShardedLines.for_engine, tests/test_gpu_distributed.py) -- the
partition and exchange code is the same."""
import os
import socket

import numpy as np
import pytest

from pylbl_amd import distributed, synthetic


def test_level_shard_partitions_every_level_once():
    for n_levels in (1, 3, 8, 64, 257):
        for world in (1, 2, 3, 8):
            covered = []
            for rank in range(world):
                s = distributed.level_shard(n_levels, rank, world)
                covered += list(range(s.start, s.stop))
            assert covered == list(range(n_levels))
            sizes = distributed.shard_sizes(n_levels, world)
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n_levels


def test_partition_covers_every_unit_once_and_fills_the_ranks():
    lines = [110_000, 400_000, 400_000, 160_000, 6_000, 300_000, 15_000, 1_000]
    for n_levels in (1, 2, 3, 7, 8, 64, 256):
        for world in (1, 2, 4, 8):
            for weights in (lines, lines[:3], lines[:1]):
                plan = distributed.partition(n_levels, weights, world)
                flat = [u for rank in range(world) for u in plan.units[rank]]
                expect = [(l, m) for l in range(n_levels) for m in range(len(weights))]
                assert flat == expect               # level-major, contiguous runs, no unit twice
                if n_levels >= world or len(weights) == 1:
                    assert plan.mode == "levels"
                    for rank in range(world):       # whole levels: sums over gases stay local
                        levels = plan.levels_of(rank)
                        assert plan.units[rank] == [(l, m) for l in levels
                                                    for m in range(len(weights))]
                else:
                    assert plan.mode == "units"
                for rank in range(world):
                    for m, levels in plan.by_molecule(rank).items():
                        assert levels == list(range(levels[0], levels[-1] + 1))
    # BASELINE configs[2]: one level x eight molecules uses every one of eight GPUs' worth of
    # weight where it can (the two 400 k-line molecules are a rank each).
    plan = distributed.partition(1, lines, 8)
    busy = [rank for rank in range(8) if plan.units[rank]]
    assert len(busy) >= 5
    heaviest = max(sum(lines[m] for _, m in plan.units[rank]) for rank in range(8))
    assert heaviest <= 400_000          # no rank carries more than the heaviest single unit
    # configs[3] / [4]: 8 and 32 whole levels per GPU.
    assert [len(distributed.partition(64, lines[:3], 8).levels_of(r)) for r in range(8)] == [8]*8
    assert [len(distributed.partition(256, lines, 8).levels_of(r)) for r in range(8)] == [32]*8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


FORMULAS = ("H2O", "CO2", "O3")
EIGHT = ("H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2", "N2")
GRID = (1, 41, 10)


def _tables(formulas=FORMULAS):
    return {f: synthetic.line_table(f, 1., 80., num_lines=60 + 40*i, tips_range=(150, 400))
            for i, f in enumerate(formulas)}


def _worker(rank, world, port, n_levels, dst, output, queue, FORMULAS=FORMULAS, members=None):
    """members: global ranks of a sub-group that does the work (the others only join the
    rendezvous); `dst` is then a rank *within that group*."""
    import torch
    import torch.distributed as dist
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    group = None
    if members is not None:
        group = dist.new_group(list(members))
        if rank not in members:
            queue.put((rank, True, 0))
            dist.barrier()
            dist.destroy_process_group()
            return
    group_rank = dist.get_rank(group)
    tables = _tables(FORMULAS)
    atmos = synthetic.standard_atmosphere(max(n_levels, 2))
    t, p = atmos.t[:n_levels], atmos.p[:n_levels]
    vmr = {f: atmos.vmr[f][:n_levels] for f in FORMULAS}
    v0, vn, npv = GRID
    calls = []

    def compute(formula, temperature, pressure, x, out, accumulate):
        calls.append((formula, len(temperature)))
        rows = np.asarray([oracle.absorption_port(tables[formula], temperature[i], pressure[i],
                                                  x[i], v0, vn, npv)[0]
                           for i in range(len(temperature))]).reshape(len(temperature), -1)
        if accumulate:
            out += torch.from_numpy(rows)
        else:
            out.copy_(torch.from_numpy(rows))

    sharded = distributed.ShardedLines(compute, FORMULAS, (vn - v0)*npv, group=group,
                                       weights=[tables[f].num_lines for f in FORMULAS])
    ok = True
    for async_op in (False, True):
        out = sharded.run(t, p, vmr, dst=dst, output=output, async_op=async_op)
        if async_op:
            out = out.wait()
        expect = {f: np.asarray([oracle.absorption_port(tables[f], t[i], p[i], vmr[f][i],
                                                        v0, vn, npv)[0] for i in range(n_levels)])
                  for f in FORMULAS}
        receives = dst is None or group_rank == dst
        if world > 1 and (members is None or len(members) > 1):
            exchange = sharded.last_exchange
            ok = ok and exchange is not None and exchange.done and exchange.seconds >= 0. \
                and group_rank not in exchange.peers
            if dst is not None:
                ok = ok and (exchange.bytes_received == 0 or receives)
        if output == "total":
            total = sum(expect[f] for f in FORMULAS)
            ok = ok and ((out is None) if not receives else
                         bool(np.allclose(out.numpy(), total, rtol=1e-14, atol=0.)))
        else:
            ok = ok and (all(v is None for v in out.values()) if not receives else
                         all(np.array_equal(out[f].numpy(), expect[f]) for f in FORMULAS))
    # Every unit was computed exactly once over the ranks (twice: two runs).
    queue.put((rank, ok, sum(levels for _, levels in calls)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_levels,dst,output", [
    (2, 4, 0, "gas"), (2, 3, 0, "gas"), (2, 5, None, "gas"), (2, 1, 0, "gas"),
    (2, 5, 1, "total"), (2, 1, 0, "total"), (2, 1, None, "total"),
    (3, 2, 0, "gas"), (3, 2, 0, "total"), (3, 1, None, "gas"),
])
def test_ranks_shard_units_and_collect(world, n_levels, dst, output):
    import torch.multiprocessing as mp
    context = mp.get_context("spawn")
    queue = context.Queue()
    port = _free_port()
    _run_ranks(world, n_levels, dst, output)


def _run_ranks(world, n_levels, dst, output, formulas=FORMULAS, members=None):
    import torch.multiprocessing as mp
    context = mp.get_context("spawn")
    queue = context.Queue()
    port = _free_port()
    procs = [context.Process(target=_worker, args=(r, world, port, n_levels, dst, output, queue,
                                                   formulas, members))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [queue.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[:2] for r in results) == [(r, True) for r in range(world)]
    assert sum(r[2] for r in results) == 2*n_levels*len(formulas)


def test_config2_units_over_eight_ranks_total_on_every_rank():
    """BASELINE configs[2] on an 8-GPU node, rehearsed on the CPU: one level x eight molecules is
    cut into (level, molecule) units, the molecules of the one level sit on several ranks, so
    the total over gases needs the one real exchange step of the path (all_reduce, dst=None)."""
    plan = distributed.partition(1, [tables.num_lines for tables in _tables(EIGHT).values()], 8)
    assert plan.mode == "units" and sum(1 for r in range(8) if plan.units[r]) >= 4
    _run_ranks(8, 1, None, "total", formulas=EIGHT)


@pytest.mark.parametrize("n_levels,dst,output", [(3, 0, "gas"), (1, 1, "total")])
def test_sub_group_addresses_its_ranks_by_global_rank(n_levels, dst, output):
    """Ranks 1 and 2 of a three-rank job form the group that computes: `dst` and the senders are
    ranks within the group, torch's send/recv/reduce want global ranks."""
    _run_ranks(3, n_levels, dst, output, members=(1, 2))


def _late_worker(rank, port, queue):
    import time
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)

    def compute(formula, temperature, pressure, x, out, accumulate):
        out.fill_(1.)
    sharded = distributed.ShardedLines(compute, ("CO2",), 64)
    t = np.asarray([250., 260.])
    if rank == 1:
        time.sleep(4.)                       # the peer that is not there in time
    started = time.perf_counter()
    try:
        pending = sharded.run(t, t*100., {"CO2": t*1e-6}, dst=0, async_op=True)
        pending.wait(timeout=1.0 if rank == 0 else 30.)
        outcome = "completed"
    except distributed.ExchangeTimeout as error:
        outcome = str(error)
    except RuntimeError as error:           # rank 1: its peer has left meanwhile
        outcome = f"peer gone: {error}"
    queue.put((rank, outcome, time.perf_counter() - started))
    queue.close()
    queue.join_thread()
    # A timed-out exchange cannot be cancelled: the process leaves (bench.py: os._exit).
    os._exit(0)


def test_exchange_timeout_names_the_rank_and_its_peers():
    import torch.multiprocessing as mp
    context = mp.get_context("spawn")
    queue = context.Queue()
    port = _free_port()
    procs = [context.Process(target=_late_worker, args=(r, port, queue)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict((r[0], r[1:]) for r in (queue.get(timeout=120) for _ in procs))
    for p in procs:
        p.join(timeout=60)
    message, seconds = results[0]
    assert "rank 0" in message and "[1]" in message and "not complete after 1 s" in message
    assert "B to receive" in message and seconds < 3.5


def _back_to_back_worker(rank, port, n_levels, output, dst, queue):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    formulas = ("H2O", "CO2", "O3")

    def compute(formula, temperature, pressure, x, out, accumulate):
        rows = torch.from_numpy(np.asarray(temperature)*1000. + formulas.index(formula))
        if accumulate:
            out += rows[:, None]
        else:
            out.copy_(rows[:, None].expand_as(out))
    sharded = distributed.ShardedLines(compute, formulas, 4096, weights=[3., 2., 1.])
    pendings, temperatures = [], []
    for call in range(4):
        t = 200. + 10.*call + np.arange(n_levels, dtype=np.float64)
        temperatures.append(t)
        # No wait between the calls: the fourth writes the per-rank blocks the second's exchange
        # read, the third those of the first.
        pendings.append(sharded.run(t, t*100., {f: t*1e-6 for f in formulas}, dst=dst,
                                    output=output, async_op=True))
    ok = True
    for t, pending in zip(temperatures, pendings):
        out = pending.wait()
        receives = dst is None or rank == dst
        if not receives:
            ok = ok and (out is None or all(v is None for v in out.values()))
        elif output == "total":
            expect = sum(t*1000. + m for m in range(3))
            ok = ok and bool(np.array_equal(out.numpy(), np.repeat(expect[:, None], 4096, axis=1)))
        else:
            for m, f in enumerate(formulas):
                ok = ok and bool(np.array_equal(out[f].numpy(),
                                                np.repeat((t*1000. + m)[:, None], 4096, axis=1)))
    queue.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_levels,output,dst", [(5, "gas", 0), (4, "total", 1), (1, "total", None),
                                                 (1, "total", 0)])
def test_calls_queued_back_to_back_without_waiting(n_levels, output, dst):
    """four asynchronous calls in a row, none waited for until all are queued -- every
    one of them must still deliver its own result (a buffer's next writer is ordered behind the
    exchange that last used it)."""
    import torch.multiprocessing as mp
    context = mp.get_context("spawn")
    queue = context.Queue()
    port = _free_port()
    procs = [context.Process(target=_back_to_back_worker,
                             args=(r, port, n_levels, output, dst, queue)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(queue.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == [(0, True), (1, True)]
