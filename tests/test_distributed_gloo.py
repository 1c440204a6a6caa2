"""The N > 1 path on CPU: two gloo ranks shard the levels and gather them.  The per-rank
compute is the CPU oracle here (test infrastructure); on GPUs it is the HIP engine
(pylbl_amd.distributed.ShardedLines.for_engine), the sharding and gather code is the same."""
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


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_levels, dst, queue):
    import torch
    import torch.distributed as dist
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tables = {f: synthetic.line_table(f, 1., 80., num_lines=60, tips_range=(150, 400))
              for f in ("H2O", "CO2")}
    atmos = synthetic.standard_atmosphere(n_levels)
    v0, vn, npv = 1, 41, 10

    def compute(formula, t, p, x):
        rows = [oracle.absorption_port(tables[formula], t[i], p[i], x[i], v0, vn, npv)[0]
                for i in range(len(t))]
        return torch.from_numpy(np.asarray(rows).reshape(len(t), (vn - v0)*npv))

    sharded = distributed.ShardedLines(compute)
    vmr = {f: atmos.vmr[f] for f in tables}
    out = sharded.run(atmos.t, atmos.p, vmr, dst=dst)
    if dst is None or rank == dst:
        expect = {f: np.asarray([oracle.absorption_port(tables[f], atmos.t[i], atmos.p[i],
                                                        atmos.vmr[f][i], v0, vn, npv)[0]
                                 for i in range(n_levels)]) for f in tables}
        ok = all(np.array_equal(out[f].numpy(), expect[f]) for f in tables)
        queue.put((rank, ok))
    else:
        queue.put((rank, all(v is None for v in out.values())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_levels,dst", [(4, 0), (3, 0), (5, None), (1, 0)])
def test_two_ranks_shard_and_gather(n_levels, dst):
    import torch.multiprocessing as mp
    context = mp.get_context("spawn")
    queue = context.Queue()
    port = _free_port()
    procs = [context.Process(target=_worker, args=(r, 2, port, n_levels, dst, queue))
             for r in range(2)]
    for p in procs:
        p.start()
    results = [queue.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]
