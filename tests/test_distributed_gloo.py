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
GRID = (1, 41, 10)


def _tables():
    return {f: synthetic.line_table(f, 1., 80., num_lines=60 + 40*i, tips_range=(150, 400))
            for i, f in enumerate(FORMULAS)}


def _worker(rank, world, port, n_levels, dst, output, queue):
    import torch
    import torch.distributed as dist
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tables = _tables()
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

    sharded = distributed.ShardedLines(compute, FORMULAS, (vn - v0)*npv,
                                       weights=[tables[f].num_lines for f in FORMULAS])
    ok = True
    for async_op in (False, True):
        out = sharded.run(t, p, vmr, dst=dst, output=output, async_op=async_op)
        if async_op:
            out = out.wait()
        expect = {f: np.asarray([oracle.absorption_port(tables[f], t[i], p[i], vmr[f][i],
                                                        v0, vn, npv)[0] for i in range(n_levels)])
                  for f in FORMULAS}
        receives = dst is None or rank == dst
        if output == "total":
            total = expect["H2O"] + expect["CO2"] + expect["O3"]
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
    procs = [context.Process(target=_worker, args=(r, world, port, n_levels, dst, output, queue))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [queue.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[:2] for r in results) == [(r, True) for r in range(world)]
    assert sum(r[2] for r in results) == 2*n_levels*len(FORMULAS)
